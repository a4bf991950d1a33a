// Measured ceilings of the box next to the spec sheet: what a kernel made of nothing but
// v_mfma_f32_32x32x2_f32 sustains (power / clock limited), and what a device-to-device copy sustains from HBM.
//   hipcc --offload-arch=gfx950 -O3 tools/ceilings/ceilings.hip -o tools/ceilings/ceilings && tools/ceilings/ceilings
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int NACC>
__global__ void __launch_bounds__(256) mfma_only(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int t = 0; t < NACC; ++t)
        for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
    }
    float s = 0.f;
    for (int t = 0; t < NACC; ++t)
        for (int q = 0; q < 16; ++q) s += acc[t][q];
    if (s == 123.456f) out[blockIdx.x * 256 + threadIdx.x] = s;      // never true: keeps the MFMAs alive
}

// same instruction stream, but operands are random data held in 16 + 4 different registers (what a real contraction
// toggles in the multipliers): shows how much of the all-ones figure is power / clock headroom
__global__ void __launch_bounds__(256) mfma_random_operands(float* out, const float* __restrict__ rnd, int iters) {
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t)
        for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
    float a[16], b[4];
    for (int i = 0; i < 16; ++i) a[i] = rnd[(threadIdx.x * 16 + i) & 4095];
    for (int i = 0; i < 4; ++i) b[i] = rnd[(threadIdx.x * 4 + i + 77) & 4095];
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * r + t], b[r], acc[t], 0, 0, 0);
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t)
        for (int q = 0; q < 16; ++q) s += acc[t][q];
    if (s == 123.456f) out[blockIdx.x * 256 + threadIdx.x] = s;
}

// The scorer's main loop in isolation (edge_score_stream_kernel<8>): 4 accumulators per wave, A operand streamed from a
// packed [4][32][64] float4 table (L2 resident), B operand = product of two gathered rows; MODE 0: A loads only (B constant),
// 1: A + B loads from ONE row, 2: A + B loads from per-lane random rows.  `tiles` main loops per wave, no epilogue.
template <int MODE>
__global__ void __launch_bounds__(256, 3) scorer_main_loop(float* out, const float4* __restrict__ Wp, const float* __restrict__ codes,
                                                          const int* __restrict__ rows, int tiles) {
    constexpr int H = 256, NTW = 4, NJ4 = 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hh = wave >> 1, kh = lane >> 5;
    const float4* wp = Wp + (static_cast<long>(hh) * NTW * NJ4) * 64 + lane;
    f32x16 acc[NTW];
    for (int t = 0; t < NTW; ++t)
        for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
    for (int tile = 0; tile < tiles; ++tile) {
        const int idx = (blockIdx.x * tiles + tile) * 256 + threadIdx.x;
        const int s = MODE == 2 ? rows[idx & 0xFFFFF] : 0, d = MODE == 2 ? rows[(idx + 4097) & 0xFFFFF] : 0;
        const float4* xp = reinterpret_cast<const float4*>(codes + static_cast<long>(s) * H + kh * (H / 2));
        const float4* yp = reinterpret_cast<const float4*>(codes + static_cast<long>(d) * H + kh * (H / 2));
        float4 A0[NTW], A1[NTW], x0, y0, x1, y1;
        x0 = y0 = x1 = y1 = make_float4(1.f, 1.f, 1.f, 1.f);
        auto load = [&](int j4, float4 (&A)[NTW], float4& x, float4& y) {
#pragma unroll
            for (int t = 0; t < NTW; ++t) A[t] = wp[(static_cast<long>(t) * NJ4 + j4) * 64];
            if (MODE >= 1) { x = xp[j4]; y = yp[j4]; }
        };
        auto mma = [&](const float4 (&A)[NTW], const float4& x, const float4& y) {
            const float b[4] = {x.x * y.x, x.y * y.y, x.z * y.z, x.w * y.w};
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int t = 0; t < NTW; ++t) {
                    const float av = jj == 0 ? A[t].x : jj == 1 ? A[t].y : jj == 2 ? A[t].z : A[t].w;
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[jj], acc[t], 0, 0, 0);
                }
        };
        load(0, A0, x0, y0);
#pragma unroll 1
        for (int j4 = 0; j4 < NJ4; j4 += 2) {
            load(j4 + 1, A1, x1, y1);
            mma(A0, x0, y0);
            if (j4 + 2 < NJ4) load(j4 + 2, A0, x0, y0);
            mma(A1, x1, y1);
        }
    }
    float sum = 0.f;
    for (int t = 0; t < NTW; ++t)
        for (int q = 0; q < 16; ++q) sum += acc[t][q];
    if (sum == 123.456f) out[blockIdx.x * 256 + threadIdx.x] = sum;
}

// Wave-tile shape study for the same contraction: EG edge groups (32 edges each) x NTW hidden tiles per wave => EG*NTW
// accumulators; per 4 k2-steps a wave loads NTW A float4 + 2*EG B float4 and issues 4*EG*NTW MFMAs.  STAGES = register stages.
template <int EG, int NTW, int WPS>
__global__ void __launch_bounds__(256, WPS) scorer_tile_shape(float* out, const float4* __restrict__ Wp, const float* __restrict__ codes,
                                                             const int* __restrict__ rows, int tiles) {
    constexpr int H = 256, NJ4 = 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, kh = lane >> 5;
    const float4* wp = Wp + lane + (wave & 1) * 64 * NJ4;
    f32x16 acc[EG][NTW];
    for (int g = 0; g < EG; ++g)
        for (int t = 0; t < NTW; ++t)
            for (int q = 0; q < 16; ++q) acc[g][t][q] = 0.f;
    for (int tile = 0; tile < tiles; ++tile) {
        const float4 *xp[EG], *yp[EG];
        for (int g = 0; g < EG; ++g) {
            const int idx = ((blockIdx.x * tiles + tile) * EG + g) * 256 + threadIdx.x;
            const int s = rows[idx & 0xFFFFF] >> 6 << 6, d = rows[(idx + 4097) & 0xFFFFF];     // src: runs of 64, dst: random
            xp[g] = reinterpret_cast<const float4*>(codes + static_cast<long>(s) * H + kh * (H / 2));
            yp[g] = reinterpret_cast<const float4*>(codes + static_cast<long>(d) * H + kh * (H / 2));
        }
        float4 A0[NTW], A1[NTW], x0[EG], y0[EG], x1[EG], y1[EG];
        auto load = [&](int j4, float4 (&A)[NTW], float4 (&x)[EG], float4 (&y)[EG]) {
#pragma unroll
            for (int t = 0; t < NTW; ++t) A[t] = wp[(static_cast<long>(t) * 2 * NJ4 + j4) * 64];
#pragma unroll
            for (int g = 0; g < EG; ++g) { x[g] = xp[g][j4]; y[g] = yp[g][j4]; }
        };
        auto mma = [&](const float4 (&A)[NTW], const float4 (&x)[EG], const float4 (&y)[EG]) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int g = 0; g < EG; ++g) {
                    const float bx = jj == 0 ? x[g].x : jj == 1 ? x[g].y : jj == 2 ? x[g].z : x[g].w;
                    const float by = jj == 0 ? y[g].x : jj == 1 ? y[g].y : jj == 2 ? y[g].z : y[g].w;
                    const float b = bx * by;
#pragma unroll
                    for (int t = 0; t < NTW; ++t) {
                        const float av = jj == 0 ? A[t].x : jj == 1 ? A[t].y : jj == 2 ? A[t].z : A[t].w;
                        acc[g][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b, acc[g][t], 0, 0, 0);
                    }
                }
        };
        load(0, A0, x0, y0);
#pragma unroll 1
        for (int j4 = 0; j4 < NJ4; j4 += 2) {
            load(j4 + 1, A1, x1, y1);
            mma(A0, x0, y0);
            if (j4 + 2 < NJ4) load(j4 + 2, A0, x0, y0);
            mma(A1, x1, y1);
        }
    }
    float sum = 0.f;
    for (int g = 0; g < EG; ++g)
        for (int t = 0; t < NTW; ++t)
            for (int q = 0; q < 16; ++q) sum += acc[g][t][q];
    if (sum == 123.456f) out[blockIdx.x * 256 + threadIdx.x] = sum;
}

__global__ void __launch_bounds__(256) mfma16_only(float* out, int iters, float a0, float b0) {
    f32x4 acc[8];
    for (int t = 0; t < 8; ++t)
        for (int q = 0; q < 4; ++q) acc[t][q] = 0.f;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
    }
    float s = 0.f;
    for (int t = 0; t < 8; ++t)
        for (int q = 0; q < 4; ++q) s += acc[t][q];
    if (s == 123.456f) out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) copy_f4(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {
    size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
    const size_t stride = static_cast<size_t>(gridDim.x) * 256;
    for (; i < n; i += stride) dst[i] = src[i];
}

template <typename F>
static float time_ms(F f, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d", p.gcnArchName, cus, p.clockRate / 1000);
    float* out;
    CK(hipMalloc(&out, 1 << 24));
    const int iters = 2000;
    for (int wg_per_cu : {1, 2, 3}) {       // 4 waves per workgroup -> 1, 2, 3 waves per SIMD
        const int grid = cus * wg_per_cu;
        float ms = time_ms([&] { hipLaunchKernelGGL((mfma_only<4>), dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 1.0f); }, 5);
        const double flops = double(grid) * 4 /*waves*/ * iters * 16.0 * (32.0 * 32 * 2 * 2);
        printf(", \"mfma_f32_32x32x2_%dwaves_per_simd_tflops\": %.1f", wg_per_cu, flops / ms / 1e9);
    }
    {
        float* rnd;
        CK(hipMalloc(&rnd, 4096 * 4));
        float h[4096];
        unsigned x = 12345u;
        for (int i = 0; i < 4096; ++i) { x = x * 1664525u + 1013904223u; h[i] = (float)(x >> 8) / 16777216.0f * 2.0f - 1.0f; }
        CK(hipMemcpy(rnd, h, sizeof(h), hipMemcpyHostToDevice));
        for (int wg_per_cu : {1, 3}) {
            const int grid = cus * wg_per_cu;
            float ms = time_ms([&] { hipLaunchKernelGGL(mfma_random_operands, dim3(grid), dim3(256), 0, 0, out, rnd, iters * 4); }, 5);
            const double flops = double(grid) * 4 * (iters * 4) * 16.0 * (32.0 * 32 * 2 * 2);
            printf(", \"mfma_f32_32x32x2_random_operands_%dwaves_per_simd_tflops\": %.1f", wg_per_cu, flops / ms / 1e9);
        }
    }
    {
        float4* Wp; float* codes; int* rows;
        const int N = 1013;
        CK(hipMalloc(&Wp, 2 * 4 * 32 * 64 * 16)); CK(hipMalloc(&codes, N * 256 * 4)); CK(hipMalloc(&rows, (1 << 20) * 4));
        CK(hipMemset(Wp, 0, 2 * 4 * 32 * 64 * 16)); CK(hipMemset(codes, 0, N * 256 * 4));
        int* hr = (int*)malloc((1 << 20) * 4);
        unsigned x = 777u;
        for (int i = 0; i < (1 << 20); ++i) { x = x * 1664525u + 1013904223u; hr[i] = (int)((x >> 8) % N); }
        CK(hipMemcpy(rows, hr, (1 << 20) * 4, hipMemcpyHostToDevice));
        const int grid = cus * 3, tiles = 8;
        const double flops = double(grid) * 4 * tiles * 512.0 * (32.0 * 32 * 2 * 2);
        float ms = time_ms([&] { hipLaunchKernelGGL((scorer_main_loop<0>), dim3(grid), dim3(256), 0, 0, out, Wp, codes, rows, tiles); }, 5);
        printf(", \"scorer_main_loop_A_stream_only_tflops\": %.1f", flops / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL((scorer_main_loop<1>), dim3(grid), dim3(256), 0, 0, out, Wp, codes, rows, tiles); }, 5);
        printf(", \"scorer_main_loop_A_plus_B_one_row_tflops\": %.1f", flops / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL((scorer_main_loop<2>), dim3(grid), dim3(256), 0, 0, out, Wp, codes, rows, tiles); }, 5);
        printf(", \"scorer_main_loop_A_plus_B_random_rows_tflops\": %.1f", flops / ms / 1e9);
        float4* Wbig;
        CK(hipMalloc(&Wbig, 8 * 2 * 32 * 64 * 16 + 4096)); CK(hipMemset(Wbig, 0, 8 * 2 * 32 * 64 * 16 + 4096));
#define SHAPE(EG, NTW, WPS, name) { const int g_ = cus * WPS; const double fl_ = double(g_) * 4 * tiles * (128.0 * EG * NTW) * (32.0 * 32 * 2 * 2); \
        float m_ = time_ms([&] { hipLaunchKernelGGL((scorer_tile_shape<EG, NTW, WPS>), dim3(g_), dim3(256), 0, 0, out, Wbig, codes, rows, tiles); }, 5); \
        printf(", \"" name "\": %.1f", fl_ / m_ / 1e9); }
        SHAPE(1, 4, 3, "shape_32e_x_128h_3wps_tflops")
        SHAPE(1, 4, 2, "shape_32e_x_128h_2wps_tflops")
        SHAPE(2, 4, 2, "shape_64e_x_128h_2wps_tflops")
        SHAPE(1, 8, 2, "shape_32e_x_256h_2wps_tflops")
        SHAPE(2, 2, 3, "shape_64e_x_64h_3wps_tflops")
        SHAPE(4, 2, 2, "shape_128e_x_64h_2wps_tflops")
        SHAPE(2, 4, 1, "shape_64e_x_128h_1wps_tflops")
        SHAPE(2, 8, 1, "shape_64e_x_256h_1wps_tflops")
        SHAPE(4, 4, 1, "shape_128e_x_128h_1wps_tflops")
        SHAPE(3, 4, 1, "shape_96e_x_128h_1wps_tflops")
    }
    {
        const int grid = cus * 2;
        float ms = time_ms([&] { hipLaunchKernelGGL(mfma16_only, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 1.0f); }, 5);
        const double flops = double(grid) * 4 * iters * 32.0 * (16.0 * 16 * 4 * 2);
        printf(", \"mfma_f32_16x16x4_2waves_per_simd_tflops\": %.1f", flops / ms / 1e9);
    }
    {
        const size_t bytes = size_t(4) << 30;     // 4 GiB each way: far beyond the 256 MiB Infinity Cache
        float4 *src, *dst;
        CK(hipMalloc(&src, bytes)); CK(hipMalloc(&dst, bytes));
        CK(hipMemset(src, 1, bytes));
        CK(hipMemset(dst, 0, bytes));
        const size_t n = bytes / 16;
        float ms = time_ms([&] { hipLaunchKernelGGL(copy_f4, dim3(cus * 16), dim3(256), 0, 0, src, dst, n); }, 5);
        printf(", \"hbm_copy_read_plus_write_GBps\": %.0f", 2.0 * bytes / ms / 1e6);
        CK(hipFree(src)); CK(hipFree(dst));
    }
    printf("}\n");
    return 0;
}
