// Weight-gradient GEMM  C[M,N] = A^T B  with A [K,M], B [K,N] row-major (K = number of graph nodes):
// dW = dY^T X of every node-level Linear on the path (model.py:94-95,151-153 GCNConv.lin, fc layers).
// The vendor library picks a 256x256 macro-tile for these skinny shapes (M = 256, N = 602, K = 1013 ->
// 3 workgroups, 233 us); here one wave owns a 32x32 output tile and a K-slice and feeds
// v_mfma_f32_32x32x2_f32 straight from global memory: the reduction index is the ROW index of both
// operands, so the A and B operand of lane l (row k + (l >> 5), column l & 31) are coalesced 128-B reads
// and no LDS / transpose is needed.  Split-K partial tiles are combined in a fixed order (deterministic).
#include "sgs_common.h"

namespace sgs {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int kUnroll = 8;

__global__ void __launch_bounds__(64) gemm_tn_tile(const float* __restrict__ A, const float* __restrict__ B, int64_t K, int M, int N,
                                                  int ksplit, float* __restrict__ slab) {
    const int lane = threadIdx.x, kh = lane >> 5, l31 = lane & 31;
    const int ti = blockIdx.x, tj = blockIdx.y, s = blockIdx.z;
    const int i = ti * 32 + l31, j = tj * 32 + l31;
    const bool iok = i < M, jok = j < N;
    const int64_t per = ((K + ksplit - 1) / ksplit + 1) & ~int64_t(1);      // even slice length
    const int64_t k0 = s * per, k1 = (k0 + per < K) ? k0 + per : K;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    int64_t k = k0;
    for (; k + 2 * kUnroll <= k1; k += 2 * kUnroll) {
        float a[kUnroll], b[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int64_t kk = k + 2 * u + kh;
            a[u] = iok ? A[kk * M + i] : 0.f;
            b[u] = jok ? B[kk * N + j] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
    }
    for (; k < k1; k += 2) {
        const int64_t kk = k + kh;
        const float a = (iok && kk < k1) ? A[kk * M + i] : 0.f;
        const float b = (jok && kk < k1) ? B[kk * N + j] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    float* out = slab + static_cast<int64_t>(s) * M * N;
    if (jok) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (row < M) out[static_cast<int64_t>(row) * N + j] = acc[r];
        }
    }
}

// Partition-sized K (K = ~1 000 graph nodes, 2 .. 8 K-slices): the slices of one 32 x 32 output tile are the NW waves of ONE
// workgroup, their partial tiles meet in LDS in a fixed tree and wave 0 writes C (row stride ldc) -- no slab round trip through
// HBM and no reduction launch (a step of the learned branch holds eight of these products; each second launch cost ~5 us).
constexpr int kWgUnroll = 8;              // k2-steps in flight per wave (16 measured slower by ~1.5 us per launch)
template <int NW>
__global__ void __launch_bounds__(64 * NW) gemm_tn_wg(const float* __restrict__ A, const float* __restrict__ B, int64_t K, int M, int N,
                                                     float* __restrict__ C, int64_t ldc) {
    __shared__ float red[NW][16][64];                                       // every wave's partial tile: 4 KB each
    const int lane = threadIdx.x & 63, s = threadIdx.x >> 6, kh = lane >> 5, l31 = lane & 31;
    const int ti = blockIdx.x, tj = blockIdx.y;
    const int i = ti * 32 + l31, j = tj * 32 + l31;
    const bool iok = i < M, jok = j < N;
    const int64_t per = ((K + NW - 1) / NW + 1) & ~int64_t(1);              // even slice length, as gemm_tn_tile with ksplit = NW
    const int64_t k0 = s * per, k1 = (k0 + per < K) ? k0 + per : K;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    int64_t k = k0;
    for (; k + 2 * kWgUnroll <= k1; k += 2 * kWgUnroll) {
        float a[kWgUnroll], b[kWgUnroll];
#pragma unroll
        for (int u = 0; u < kWgUnroll; ++u) {
            const int64_t kk = k + 2 * u + kh;
            a[u] = iok ? A[kk * M + i] : 0.f;
            b[u] = jok ? B[kk * N + j] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < kWgUnroll; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
    }
    for (; k < k1; k += 2) {
        const int64_t kk = k + kh;
        const float a = (iok && kk < k1) ? A[kk * M + i] : 0.f;
        const float b = (jok && kk < k1) ? B[kk * N + j] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    // ONE barrier: every wave parks its tile, then wave s sums accumulator registers s * 16/NW .. of all slices in slice order (fixed) and
    // writes those rows of C -- the combine and the store are spread over the workgroup (a halving tree cost 2 log2(NW) barriers and left
    // the store to wave 0)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[s][r][lane] = acc[r];
    __syncthreads();
    if (!jok) return;
    constexpr int RPW = 16 / NW;                                            // accumulator registers finished by a wave
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
        const int r = s * RPW + q;
        float sum = 0.f;
#pragma unroll
        for (int z = 0; z < NW; ++z) sum += red[z][r][lane];
        const int row = ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (row < M) C[static_cast<int64_t>(row) * ldc + j] = sum;
    }
}

// `N`, `ldc`: C is written with row stride ldc (>= N): the result may be a column block of a wider matrix (fc1.weight's halves)
__global__ void __launch_bounds__(256) gemm_tn_reduce(const float* __restrict__ slab, int64_t mn, int ksplit, float* __restrict__ C,
                                                      const float* __restrict__ cpart = nullptr, int M = 0, float* __restrict__ colsum = nullptr,
                                                      int N = 0, int64_t ldc = 0, const float* __restrict__ rowscale = nullptr,
                                                      float scale = 1.f, const float* __restrict__ dzpart = nullptr, int nz = 0,
                                                      float* __restrict__ dzsum = nullptr, float* __restrict__ Craw = nullptr,
                                                      float* __restrict__ csraw = nullptr, int csplit = 0) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= mn) {
        const int64_t c = i - mn;                       // trailing threads: the column sums of A, slices in the same fixed order
        if (dzsum && c == M) {                          // one more: the sum of the mask operand's row factors (d fc2.bias), slice by slice
            float acc = 0.f;
            for (int z = 0; z < nz; ++z) acc += dzpart[z];
            dzsum[0] = acc;
        }
        if (colsum && c < M) {
            float acc = 0.f;
            const int ncs = csplit > 0 ? csplit : ksplit;     // (the shared-operand kernel writes NG column-sum rows per slab)
            if (csplit > 0) {
                // many rows: eight loads in flight, eight running sums joined in a fixed tree (one dependent load per row was 50 us at 256 rows)
                float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                int s = 0;
                for (; s + 8 <= ncs; s += 8) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = cpart[static_cast<int64_t>(s + j) * M + c];
#pragma unroll
                    for (int j = 0; j < 8; ++j) a8[j] += v[j];
                }
                for (; s < ncs; ++s) a8[0] += cpart[static_cast<int64_t>(s) * M + c];
                acc = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
            } else
            for (int s = 0; s < ncs; ++s) acc += cpart[static_cast<int64_t>(s) * M + c];
            if (csraw) csraw[c] = acc;                  // (the sums before the row factor: a term of d fc2.weight)
            colsum[c] = rowscale ? acc * (rowscale[c] * scale) : acc;
        }
        return;
    }
    float acc = 0.f;
    if (csplit > 0) {
        // the shared-operand kernel's slabs: eight loads in flight, eight running sums joined in a fixed tree
        float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int s = 0;
        for (; s + 8 <= ksplit; s += 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = slab[static_cast<int64_t>(s + j) * mn + i];
#pragma unroll
            for (int j = 0; j < 8; ++j) a8[j] += v[j];
        }
        for (; s < ksplit; ++s) a8[0] += slab[static_cast<int64_t>(s) * mn + i];
        acc = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
    } else
    for (int s = 0; s < ksplit; ++s) acc += slab[static_cast<int64_t>(s) * mn + i];
    if (Craw) Craw[i] = acc;                             // dense [M, N], before the row factor
    if (rowscale) acc *= rowscale[i / N] * scale;        // mask-operand products: row m of C carries the factor left out of A
    if (ldc > 0 && ldc != N) C[(i / N) * ldc + (i % N)] = acc;
    else C[i] = acc;
}

// ---------------------------------------------------------------------------------------------
// Tall-K variant (K >= 8192): the weight gradient of the scorer's fc1, dW1a = dv^T feat with K = q = 100 000 rows and
// M = N = 256 (training_hybrid.py's learned backward).  The vendor GEMM runs this shape at 42 TFLOP/s (310 us) and the
// one-tile-per-wave kernel above needs two loads per MFMA.  Here a wave owns a 128 x 64 output tile (8 accumulators) and a
// K-slice: per k2-step ONE 16-byte load of A (row k, columns m0 + 4 l31 .. + 3 -> the A operands of four M-tiles whose rows
// interleave with stride 4) and ONE 8-byte load of B (two N-tiles) feed 8 MFMAs, i.e. 0.25 loads / 24 B per lane per 8
// MFMAs, all of it coalesced row segments.  Sixteen k2-steps are in flight per wave (two register stages of eight), enough to
// cover L2 latency with a single wave per SIMD, so K is split only as far as needed to give every SIMD one wave.
constexpr int kTallK = 8192;
constexpr int kTallU = 8;                    // k2-steps per register stage

__global__ void __launch_bounds__(64) gemm_tn_tall_tile(const float* __restrict__ A, const float* __restrict__ B, int64_t K, int M, int N,
                                                       int ksplit, float* __restrict__ slab, float* __restrict__ cpart) {
    const int lane = threadIdx.x, kh = lane >> 5, l31 = lane & 31;
    const int m0 = blockIdx.x * 128, n0 = blockIdx.y * 64, s = blockIdx.z;
    const int ia = m0 + 4 * l31, jb = n0 + 2 * l31;
    const bool aok = ia < M, bok = jb < N;         // M % 4 == 0 and N % 2 == 0 (checked by the launcher): whole vectors are in or out
    const int64_t per = ((K + ksplit - 1) / ksplit + 1) & ~int64_t(1);      // even slice length
    const int64_t k0 = s * per, k1 = (k0 + per < K) ? k0 + per : K;
    f32x16 acc[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float2 z2 = make_float2(0.f, 0.f);
    // optional by-product: the column sums of A (the bias gradient d b1 = colsum(dv) when A = dv): the first N-tile's
    // waves already stream every row of their 128 columns, so the sums cost four adds per k2-step
    const bool want_cs = cpart != nullptr && blockIdx.y == 0;
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
    auto load = [&](int64_t k, float4 (&a)[kTallU], float2 (&b)[kTallU]) {
#pragma unroll
        for (int u = 0; u < kTallU; ++u) {
            const int64_t kk = k + 2 * u + kh;
            const bool in = kk < k1;
            a[u] = (aok && in) ? *reinterpret_cast<const float4*>(A + kk * M + ia) : z4;
            b[u] = (bok && in) ? *reinterpret_cast<const float2*>(B + kk * N + jb) : z2;
        }
    };
    auto mma = [&](const float4 (&a)[kTallU], const float2 (&b)[kTallU]) {
#pragma unroll
        for (int u = 0; u < kTallU; ++u) {
            const float av[4] = {a[u].x, a[u].y, a[u].z, a[u].w};
            const float bv[2] = {b[u].x, b[u].y};
            if (want_cs) { cs[0] += av[0]; cs[1] += av[1]; cs[2] += av[2]; cs[3] += av[3]; }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int w = 0; w < 2; ++w) acc[t][w] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[w], acc[t][w], 0, 0, 0);
        }
    };
    float4 a0[kTallU], a1[kTallU];
    float2 b0[kTallU], b1[kTallU];
    load(k0, a0, b0);
#pragma unroll 1
    for (int64_t k = k0; k < k1; k += 4 * kTallU) {
        load(k + 2 * kTallU, a1, b1);                       // rows past k1 load zeros
        mma(a0, b0);
        load(k + 4 * kTallU, a0, b0);
        mma(a1, b1);
    }
    if (want_cs) {
#pragma unroll
        for (int t = 0; t < 4; ++t) cs[t] += __shfl_xor(cs[t], 32, 64);          // even + odd rows of the slice
        if (kh == 0 && aok) *reinterpret_cast<float4*>(cpart + static_cast<int64_t>(s) * M + ia) = make_float4(cs[0], cs[1], cs[2], cs[3]);
    }
    float* out = slab + static_cast<int64_t>(s) * M * N;
    // accumulator register r of tile (t, w): row m0 + 4 ((r & 3) + 8 (r >> 2) + 4 kh) + t, column n0 + 2 l31 + w
    if (bok) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + 4 * ((r & 3) + 8 * (r >> 2) + 4 * kh) + t;
                if (row < M) *reinterpret_cast<float2*>(out + static_cast<int64_t>(row) * N + jb) = make_float2(acc[t][0][r], acc[t][1][r]);
            }
    }
}

// ---------------------------------------------------------------------------------------------
// The tall-K product on the bf16 matrix pipe (bf16x6, see sgs_common.h / edge_score.hip): same 128 x 64 wave tile, same
// interleaved row / column assignment and the same split-K slabs as gemm_tn_tall_tile, but a step is 16 rows of K: lane
// (l31, g) loads rows k + 8 g + j (j = 0..7) -- eight 16-byte pieces of A (its element of four M-tiles each) and eight 8-byte
// pieces of B (two N-tiles) -- splits the 48 values into three bf16 pieces each and issues 8 tiles x 6 products of
// v_mfma_f32_32x32x16_bf16.  fp32-faithful like the scorer loop; the VALU work (264 instructions per 48 MFMAs) is at the
// budget the matrix pipe leaves, so the kernel runs one wave per SIMD with the next step's loads in flight.
// NW waves per workgroup, each with its own K-slice (ksplit = NW x gridDim.z slices in all): the NW partial tiles are summed
// through LDS in a fixed tree before ONE slab per workgroup is written, so the slab reduction that follows reads 1 / NW of the
// bytes (K = 100 000, M = N = 256: 128 slices -> 32 slabs of 256 KB; the reduction launch drops from 32 us to the launch floor).
// MASK: A is a 0 / 1 matrix given as bits (Abits [K, M/32], bit m of row k) times a per-row factor dz[k]: the scorer's
// dv = diag(dz) mask diag(w2 / (1 - p)).  The mask is ONE exact bf16 piece and dz moves to the B side (b <- dz[k] * b before the split), so
// a product is 3 MFMAs instead of 6 and the A stream is 1/32 of the bytes; the column factor w2 / (1 - p) is applied by gemm_tn_reduce.
// GATHER (with MASK): B is never materialised -- row k of B is codes[src k, :] * codes[dst k, :] with (src, dst) = sd[k] (the scorer's
// feat = x_s * x_d; `B` is then the codes table [*, N]).  The endpoints of a step are fetched one step before its rows (a dependent
// gather needs its index first), the rows two steps before their products -- same values, same order of operations as reading a
// materialised feat, so C is bit-identical; what goes away is the [K, N] array's round trip through HBM (written by the prep pass,
// read once per 128-row M-tile here).
template <int NW, bool MASK = false, bool GATHER = false>
__global__ void __launch_bounds__(64 * NW) gemm_tn_tall_bf16x6(const float* __restrict__ A, const float* __restrict__ B, int64_t K, int M, int N,
                                                              int ksplit, float* __restrict__ slab, float* __restrict__ cpart,
                                                              const uint32_t* __restrict__ Abits = nullptr, const float* __restrict__ dz = nullptr,
                                                              float* __restrict__ dzpart = nullptr, const int32_t* __restrict__ sd = nullptr) {
    static_assert(!GATHER || MASK, "the gathered B operand comes with the mask form of A");
    extern __shared__ float red_lds[];       // NW > 1: [NW / 2][8 tiles x 16 registers][64 lanes] partial tiles + [NW / 2][4][64] column sums
    const int lane = threadIdx.x & 63, g = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // wave-uniform, and the compiler should know: slice bounds in SGPRs
    const int m0 = blockIdx.x * 128, n0 = blockIdx.y * 64, s = blockIdx.z * NW + wave;
    const int ia = m0 + 4 * l31, jb = n0 + 2 * l31;
    const bool aok = ia < M, bok = jb < N;
    const int64_t per = ((K + ksplit - 1) / ksplit + 15) & ~int64_t(15);   // slice length: a whole number of 16-row steps
    const int64_t k0 = s * per, k1 = (k0 + per < K) ? k0 + per : K;
    f32x16 acc[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float2 z2 = make_float2(0.f, 0.f);
    const bool want_cs = cpart != nullptr && blockIdx.y == 0;
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
    const bool want_dz = MASK && dzpart != nullptr && blockIdx.x == 0 && blockIdx.y == 0;      // the slice's sum of dz (d fc2.bias) on the way
    float dzs = 0.f;
    struct Raw { float4 a[8]; float2 b[8]; float2 b2[8]; uint32_t aw[8]; float dzr[8]; int64_t k; };      // (a | aw, dzr, b2: the unused ones are never live)
    struct Idx { int2 e[8]; };                                              // GATHER: the endpoints of one step's eight rows
    const int wsel = ia >> 5, wsh = ia & 31;
    // MASK: per-lane bases of the slice, so that a step's 24 loads are base + (row-in-slice) * stride in 32-bit arithmetic (64-bit index
    // math per load made this loop VALU-bound: ~620 vector instructions per 24 MFMAs, measured 92 us; the matrix work is 16 us)
    const int wpr = M >> 5;                                                 // mask words per row
    const int lim = static_cast<int>(k1 - k0);                              // rows in this slice (<= 0: an empty trailing slice)
    const uint32_t* Ab = MASK ? Abits + k0 * wpr + wsel : nullptr;
    const float* Db = MASK ? dz + k0 : nullptr;
    const float* Bb = GATHER ? B + jb : B + k0 * N + jb;
    const int2* Sb = GATHER ? reinterpret_cast<const int2*>(sd) + k0 : nullptr;
    // 32-bit offsets from per-lane bases throughout (the codes table has < 2^31 elements: checked by the launcher); whole steps -- all
    // but a slice's last -- take the unclamped path (wave-uniform)
    auto load_idx = [&](int64_t k, Idx& ix) {                               // (rows past the slice: its last row, neutralised through dz = 0)
        const int rel = static_cast<int>(k - k0) + 8 * g;
        if (k + 16 <= k1) {
            const int2* sp = Sb + rel;
#pragma unroll
            for (int j = 0; j < 8; ++j) ix.e[j] = sp[j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) ix.e[j] = Sb[max(min(rel + j, lim - 1), 0)];
        }
    };
    auto load_rows = [&](int64_t k, Raw& r, const Idx& ix) {
        r.k = k;
        const int rel = static_cast<int>(k - k0) + 8 * g;
        if (k + 16 <= k1) {
            const uint32_t* ap = Ab + rel * wpr;
            const float* dp = Db + rel;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                r.aw[j] = ap[j * wpr];
                r.dzr[j] = dp[j];
                r.b[j] = *reinterpret_cast<const float2*>(Bb + static_cast<uint32_t>(ix.e[j].x) * static_cast<uint32_t>(N));
                r.b2[j] = *reinterpret_cast<const float2*>(Bb + static_cast<uint32_t>(ix.e[j].y) * static_cast<uint32_t>(N));
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int rc = max(min(rel + j, lim - 1), 0);
                r.aw[j] = Ab[rc * wpr];
                r.dzr[j] = Db[rc];
                r.b[j] = *reinterpret_cast<const float2*>(Bb + static_cast<uint32_t>(ix.e[j].x) * static_cast<uint32_t>(N));
                r.b2[j] = *reinterpret_cast<const float2*>(Bb + static_cast<uint32_t>(ix.e[j].y) * static_cast<uint32_t>(N));
            }
        }
    };
    auto load = [&](int64_t k, Raw& r) {
        r.k = k;
        if constexpr (MASK) {
            // UNCONDITIONAL loads, raw values only (a load under `in ? .. : 0` with arithmetic on its result compiles to a branch with a
            // vmcnt(0) wait inside: eight serialised round trips per step, 2.1 x slower).  Whole steps (all but a slice's last) need no
            // clamping at all; rows past the slice read its last row and are neutralised in mma() through dz = 0.
            const int rel = static_cast<int>(k - k0) + 8 * g;
            if (k + 16 <= k1) {                                             // (wave-uniform)
                const uint32_t* a = Ab + rel * wpr;
                const float* d = Db + rel;
                const float* b = Bb + static_cast<int64_t>(rel) * N;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    r.aw[j] = a[j * wpr];
                    r.dzr[j] = d[j];
                    r.b[j] = *reinterpret_cast<const float2*>(b + j * N);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int rc = min(rel + j, lim - 1);
                    r.aw[j] = Ab[rc * wpr];
                    r.dzr[j] = Db[rc];
                    r.b[j] = *reinterpret_cast<const float2*>(Bb + static_cast<int64_t>(rc) * N);
                }
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t kk = k + 8 * g + j;
            const bool in = kk < k1;
            r.a[j] = (aok && in) ? *reinterpret_cast<const float4*>(A + kk * M + ia) : z4;
            r.b[j] = (bok && in) ? *reinterpret_cast<const float2*>(B + kk * N + jb) : z2;
        }
    };
    auto mma = [&](const Raw& r_) {
        Raw r = r_;
        if constexpr (GATHER) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { r.b[j].x *= r.b2[j].x; r.b[j].y *= r.b2[j].y; }      // feat = x_s * x_d, rounded once as the stored array was
        }
        if constexpr (MASK) {
#pragma unroll
            for (int j = 0; j < 8; ++j) r.aw[j] >>= wsh;            // bits of columns ia .. ia + 3 in the low nibble
            if (r.k + 16 > k1) {                                    // (wave-uniform) the slice's last step: rows past it count for nothing
                const int rel = static_cast<int>(r.k - k0) + 8 * g;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (rel + j >= lim) r.dzr[j] = 0.f;
            }
            if (want_dz) {
#pragma unroll
                for (int j = 0; j < 8; ++j) dzs += r.dzr[j];
            }
        }
        if (want_cs) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if constexpr (MASK) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) cs[t] += (r.aw[j] >> t) & 1u ? r.dzr[j] : 0.f;
                } else {
                    cs[0] += r.a[j].x; cs[1] += r.a[j].y; cs[2] += r.a[j].z; cs[3] += r.a[j].w;
                }
            }
        }
        u32x4 Bp[2][3];
#pragma unroll
        for (int w = 0; w < 2; ++w)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                float e0 = w == 0 ? r.b[2 * m].x : r.b[2 * m].y, e1 = w == 0 ? r.b[2 * m + 1].x : r.b[2 * m + 1].y;
                if constexpr (MASK) { e0 *= r.dzr[2 * m]; e1 *= r.dzr[2 * m + 1]; }
                uint32_t p1, p2, p3;
                split3(e0, e1, p1, p2, p3);
                Bp[w][0][m] = p1; Bp[w][1][m] = p2; Bp[w][2][m] = p3;
            }
        if constexpr (MASK) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                u32x4 A1;
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    A1[m] = ((r.aw[2 * m] >> t) & 1u) * 0x3F80u + ((r.aw[2 * m + 1] >> t) & 1u) * 0x3F800000u;      // bf16 1.0 / 0.0 pairs
                const bf16x8 a1 = __builtin_bit_cast(bf16x8, A1);
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    const bf16x8 b1 = __builtin_bit_cast(bf16x8, Bp[w][0]), b2 = __builtin_bit_cast(bf16x8, Bp[w][1]), b3 = __builtin_bit_cast(bf16x8, Bp[w][2]);
                    acc[t][w] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc[t][w], 0, 0, 0);     // smallest terms first
                    acc[t][w] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc[t][w], 0, 0, 0);
                    acc[t][w] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[t][w], 0, 0, 0);
                }
            }
            return;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            u32x4 Ap[3];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const float4 lo = r.a[2 * m], hi = r.a[2 * m + 1];
                const float e0 = t == 0 ? lo.x : t == 1 ? lo.y : t == 2 ? lo.z : lo.w;
                const float e1 = t == 0 ? hi.x : t == 1 ? hi.y : t == 2 ? hi.z : hi.w;
                uint32_t p1, p2, p3;
                split3(e0, e1, p1, p2, p3);
                Ap[0][m] = p1; Ap[1][m] = p2; Ap[2][m] = p3;
            }
            const bf16x8 a1 = __builtin_bit_cast(bf16x8, Ap[0]), a2 = __builtin_bit_cast(bf16x8, Ap[1]), a3 = __builtin_bit_cast(bf16x8, Ap[2]);
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                const bf16x8 b1 = __builtin_bit_cast(bf16x8, Bp[w][0]), b2 = __builtin_bit_cast(bf16x8, Bp[w][1]), b3 = __builtin_bit_cast(bf16x8, Bp[w][2]);
                acc[t][w] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc[t][w], 0, 0, 0);     // smallest terms first
                acc[t][w] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc[t][w], 0, 0, 0);
                acc[t][w] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc[t][w], 0, 0, 0);
                acc[t][w] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc[t][w], 0, 0, 0);
                acc[t][w] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc[t][w], 0, 0, 0);
                acc[t][w] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[t][w], 0, 0, 0);
            }
        }
    };
    Raw r0, r1;
    if constexpr (GATHER) {
        if (lim > 0) {
            // step n multiplies stage n % 3 while the rows of step n + 2 and the endpoints of step n + 3 are in flight
            Raw r2;
            Idx ix;
            load_idx(k0, ix);
            load_rows(k0, r0, ix);
            load_idx(k0 + 16, ix);
            load_rows(k0 + 16, r1, ix);
            load_idx(k0 + 32, ix);
#pragma unroll 1
            for (int64_t k = k0; k < k1; k += 48) {
                load_rows(k + 32, r2, ix);
                load_idx(k + 48, ix);
                mma(r0);
                load_rows(k + 48, r0, ix);
                load_idx(k + 64, ix);
                mma(r1);
                load_rows(k + 64, r1, ix);
                load_idx(k + 80, ix);
                mma(r2);
            }
        }
    } else if constexpr (!MASK) {
        load(k0, r0);
    } else {
        if (lim > 0) load(k0, r0);
    }
    if constexpr (MASK && !GATHER) {
        if (lim > 0) {
        // a stage is 32 registers here (against 48), so THREE are kept: two steps of loads in flight behind the one being multiplied --
        // with one wave per SIMD the loop runs at the memory latency per step, and the second stage in flight halves it (measured)
        Raw r2;
        load(k0 + 16, r1);
#pragma unroll 1
        for (int64_t k = k0; k < k1; k += 48) {
            load(k + 32, r2);                               // rows past k1: a clamped row with dz = 0
            mma(r0);
            load(k + 48, r0);
            mma(r1);
            load(k + 64, r1);
            mma(r2);
        }
        }
    } else if constexpr (!MASK) {
#pragma unroll 1
        for (int64_t k = k0; k < k1; k += 32) {
            load(k + 16, r1);                                   // rows past k1 load zeros
            mma(r0);
            load(k + 32, r0);
            mma(r1);
        }
    }
    if (want_cs) {
#pragma unroll
        for (int t = 0; t < 4; ++t) cs[t] += __shfl_xor(cs[t], 32, 64);          // the two 8-row groups of every step
    }
    if (want_dz) {
        dzs += __shfl_xor(dzs, 32, 64);
        if (lane == 0) dzpart[s] = dzs;                                          // one entry per K-slice; gemm_tn_reduce adds them in order
    }
    if (NW > 1) {
        // fixed tree over the workgroup's waves: upper half stores, lower half adds, halve, repeat (lane-major: conflict-free)
        float* csl = red_lds + (NW / 2) * 128 * 64;
#pragma unroll
        for (int half = NW / 2; half >= 1; half >>= 1) {
            if (wave >= half && wave < 2 * half) {
                float* dstp = red_lds + static_cast<size_t>(wave - half) * 128 * 64;
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int r = 0; r < 16; ++r) dstp[((t * 2 + u) * 16 + r) * 64 + lane] = acc[t][u][r];
#pragma unroll
                for (int t = 0; t < 4; ++t) csl[((wave - half) * 4 + t) * 64 + lane] = cs[t];
            }
            __syncthreads();
            if (wave < half) {
                const float* srcp = red_lds + static_cast<size_t>(wave) * 128 * 64;
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[t][u][r] += srcp[((t * 2 + u) * 16 + r) * 64 + lane];
#pragma unroll
                for (int t = 0; t < 4; ++t) cs[t] += csl[(wave * 4 + t) * 64 + lane];
            }
            __syncthreads();
        }
        if (wave != 0) return;
    }
    const int slab_id = blockIdx.z;
    if (want_cs && g == 0 && aok) *reinterpret_cast<float4*>(cpart + static_cast<int64_t>(slab_id) * M + ia) = make_float4(cs[0], cs[1], cs[2], cs[3]);
    float* out = slab + static_cast<int64_t>(slab_id) * M * N;
    // accumulator register r of tile (t, w): row m0 + 4 ((r & 3) + 8 (r >> 2) + 4 g) + t, column n0 + 2 l31 + w
    if (bok) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + 4 * ((r & 3) + 8 * (r >> 2) + 4 * g) + t;
                if (row < M) *reinterpret_cast<float2*>(out + static_cast<int64_t>(row) * N + jb) = make_float2(acc[t][0][r], acc[t][1][r]);
            }
    }
}

// ---------------------------------------------------------------------------------------------
// The scorer's weight gradient, shared-operand form (round 3):  T[h, c] = sum_k dz[k] mask[k, h] (codes[src k, c] codes[dst k, c]).
// gemm_tn_tall_bf16x6<.., true, true> gives every WAVE its own K-slice and a 128 x 64 tile, so every wave gathers, multiplies and splits
// all 16 rows x 64 columns of a step itself (~300 vector instructions and 40 loads per 24 MFMAs, one wave per SIMD: 84 us at K = 100 000
// where the matrix work is 16 us; counters: VALU-bound at 55 % activity, waiting on its gathers the rest of the time).  Here the four waves
// of a K-group walk the SAME rows and share the operands through LDS:
//   * wave w gathers / multiplies / splits rows 8 g + 2 w, + 1 of the step only (8 loads, two split3 per lane) and writes its dword of
//     every B fragment ([u][piece][lane][m = w]); all four read whole fragments back (six ds_read_b128 per step);
//   * the step's 16 x M / 32 mask words and 16 dz go through LDS once (coalesced), an A fragment is two broadcast ds_read_b128 + 11 vector
//     instructions per tile (bit pairs of two rows multiplied by 0x3F80 as one u24 product);
//   * a wave owns 64 (M = 256) or 32 (M = 128) hidden units x 64 columns: 2 MT accumulator tiles, 6 MT MFMAs per step against ~100 vector
//     instructions -- balanced, at half the registers, so two or three waves share a SIMD and cover each other's gathers.
// Two K-groups per workgroup (8 waves) on adjacent halves of the workgroup's K-slice, summed through LDS at the end: one slab per
// workgroup.  Three LDS stages, ONE barrier per step: step i + 1 is written while step i is multiplied; the endpoints of step i + 3 and
// the rows of step i + 2 are in flight.  Column sums of dz * mask (d b1) are split over the NG = N / 64 column groups (rows 8 / NG each).
// Fixed order throughout: run-to-run deterministic; NOT bit-identical to the per-wave-slice kernel (another summation order).
template <int MT, int NG>
__global__ void __launch_bounds__(512) gemm_tn_maskfeat(const uint32_t* __restrict__ Abits, const float* __restrict__ dz,
                                                        const float* __restrict__ codes, const int2* __restrict__ sd, int64_t K, int N,
                                                        float* __restrict__ slab, float* __restrict__ cpart, float* __restrict__ dzpart) {
    constexpr int M = 128 * MT, WPR = M / 32;
    constexpr int kBst = 2 * 3 * 64 * 4;                    // dwords of B fragments per stage: [u][piece][lane][m]
    constexpr int kMst = WPR * 16;                          // mask words per stage: [word][row]
    constexpr int kStage = kBst + kMst + 16;                // + 16 dz
    extern __shared__ uint32_t lds[];                       // [2 K-groups][3 stages][kStage]; reused for the K-group sum
    const int lane = threadIdx.x & 63, g = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kg = wave >> 2, w = wave & 3;
    const int ng = blockIdx.x, n0 = ng * 64, slab_id = blockIdx.y;             // (gridDim.x == NG)
    const int64_t per = (((K + gridDim.y - 1) / gridDim.y) + 63) & ~int64_t(63);        // workgroup slice: two halves of an EVEN number of whole steps
    const int64_t kb = static_cast<int64_t>(slab_id) * per + kg * (per >> 1);
    const int64_t ke = (kb + (per >> 1) < K) ? kb + (per >> 1) : K;                        // (ke <= kb: an empty trailing half)
    const int nsteps = static_cast<int>(per >> 5);                                          // the same for every K-group (barriers)
    const int lim = static_cast<int>(ke - kb);
    uint32_t* st0 = lds + kg * (3 * kStage);
    f32x16 acc[MT][2];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;
    float cs[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) cs[t] = 0.f;
    float dzs = 0.f;
    const bool want_cs = cpart != nullptr, want_dz = dzpart != nullptr && ng == 0;
    constexpr int rpg = 8 / NG;                              // rows (of every 8) whose dz * mask this column group sums; NG in {1, 2, 4, 8}
    // this lane's two rows of a step: 8 g + 2 w (+ 1); its staging duty: mask word (tid_g % WPR) of row (tid_g / WPR), dz of row lane
    const int tid_g = threadIdx.x & 255;
    const int srow = (tid_g / WPR) & 15, sword = tid_g % WPR;
    const bool stage_mask = tid_g < 16 * WPR, stage_dz = w == 3 && lane < 16;
    // Every load is UNCONDITIONAL on a clamped row (a load under a branch, or arithmetic on its result next to it, makes the compiler wait
    // for it on the spot -- with the dependent gather behind it that was two exposed memory round trips per step); rows past the half are
    // neutralised when the step is published, through dz = 0.  Addresses are a wave-uniform base (the half's first row) plus a 32-bit
    // offset: scalar-base loads, no 64-bit vector address arithmetic.
    struct S1 { int2 ea, eb; float da, db; uint32_t mw; float dl; };
    struct S2 { float2 as, ad, bs, bd; float da, db; uint32_t mw; float dl; };
    const int64_t kbase = lim > 0 ? kb : 0;                   // (an empty half reads row 0, with dz = 0)
    const int2* sdb = sd + kbase;
    const float* dzb = dz + kbase;
    const uint32_t* abb = Abits + kbase * WPR;
    const int last = lim > 0 ? lim - 1 : 0;
    auto clampr = [&](int rel) { return static_cast<uint32_t>(max(min(rel, last), 0)); };
    const uint32_t coff = static_cast<uint32_t>(n0 + 2 * l31);
    auto load_idx = [&](int step) {
        S1 o;
        const int rel = 16 * step + 8 * g + 2 * w;
        const uint32_t ra = clampr(rel), rb = clampr(rel + 1);
        o.ea = sdb[ra]; o.eb = sdb[rb];
        o.da = dzb[ra]; o.db = dzb[rb];
        o.mw = abb[clampr(16 * step + srow) * WPR + sword];
        o.dl = dzb[clampr(16 * step + (lane & 15))];
        return o;
    };
    auto load_rows = [&](const S1& i) {
        S2 o;
        const uint32_t un = static_cast<uint32_t>(N);
        o.as = *reinterpret_cast<const float2*>(codes + (static_cast<uint32_t>(i.ea.x) * un + coff));
        o.ad = *reinterpret_cast<const float2*>(codes + (static_cast<uint32_t>(i.ea.y) * un + coff));
        o.bs = *reinterpret_cast<const float2*>(codes + (static_cast<uint32_t>(i.eb.x) * un + coff));
        o.bd = *reinterpret_cast<const float2*>(codes + (static_cast<uint32_t>(i.eb.y) * un + coff));
        o.da = i.da; o.db = i.db; o.mw = i.mw; o.dl = i.dl;
        return o;
    };
    auto publish = [&](const S2& r, int step, uint32_t* stg) {
        const int rel = 16 * step + 8 * g + 2 * w;
        const float da = rel < lim ? r.da : 0.f, db = rel + 1 < lim ? r.db : 0.f;          // rows past the half count for nothing
        // feat = x_s * x_d rounded once, then the row factor, then the exact three-way split: as the materialised form did
        const float a0 = (r.as.x * r.ad.x) * da, a1 = (r.as.y * r.ad.y) * da;
        const float b0 = (r.bs.x * r.bd.x) * db, b1 = (r.bs.y * r.bd.y) * db;
        uint32_t p1, p2, p3;
        split3(a0, b0, p1, p2, p3);                          // column n0 + 2 l31     (u = 0): rows (8 g + 2 w, + 1) = dword m = w of its fragments
        stg[((0 * 3 + 0) * 64 + lane) * 4 + w] = p1; stg[((0 * 3 + 1) * 64 + lane) * 4 + w] = p2; stg[((0 * 3 + 2) * 64 + lane) * 4 + w] = p3;
        split3(a1, b1, p1, p2, p3);                          // column n0 + 2 l31 + 1 (u = 1)
        stg[((1 * 3 + 0) * 64 + lane) * 4 + w] = p1; stg[((1 * 3 + 1) * 64 + lane) * 4 + w] = p2; stg[((1 * 3 + 2) * 64 + lane) * 4 + w] = p3;
        if (stage_mask) stg[kBst + sword * 16 + srow] = r.mw;
        if (stage_dz) stg[kBst + kMst + lane] = __float_as_uint(16 * step + lane < lim ? r.dl : 0.f);
        if (want_dz && l31 == 0) dzs += da + db;             // every row of the step exactly once over (w, g)
    };
    // hidden units of this wave: h = 32 MT w + MT l31 + t (interleaved: one mask word serves both tiles)
    const int wsel = MT * w + ((MT * l31) >> 5), wsh = (MT * l31) & 31;
    // A step's operands: read from LDS and expanded after the barrier, with the MFMAs.  (Measured at K = 100 000: reading them BEFORE the
    // barrier, behind the next step's publish arithmetic, 88 us; reading and expanding them there, 104 us; all of it after the barrier,
    // 81 us -- the half-step offset between the K-groups wants about half of a wave's work on either side of the barrier.)
    struct Ops { bf16x8 bq[2][3]; uint4 m0, m1; uint32_t cw[8 / NG]; float cd[8 / NG]; };
    auto fetch = [&](const uint32_t* stg, Ops& o) {
        o.m0 = *reinterpret_cast<const uint4*>(stg + kBst + wsel * 16 + 8 * g);
        o.m1 = *reinterpret_cast<const uint4*>(stg + kBst + wsel * 16 + 8 * g + 4);
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc)
                o.bq[u][pc] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(stg + ((u * 3 + pc) * 64 + lane) * 4));
        if (want_cs) {
            // this column group's rows of every 8 (rpg of them): their mask word and dz straight from LDS (a runtime row index into
            // registers would put them in scratch)
#pragma unroll
            for (int jj = 0; jj < rpg; ++jj) {
                const int j = 8 * g + ng * rpg + jj;
                o.cw[jj] = stg[kBst + wsel * 16 + j];
                o.cd[jj] = __uint_as_float(stg[kBst + kMst + j]);
            }
        }
    };
    auto mma = [&](const Ops& o) {
        uint32_t aw[8] = {o.m0.x, o.m0.y, o.m0.z, o.m0.w, o.m1.x, o.m1.y, o.m1.z, o.m1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) aw[j] >>= wsh;
        uint32_t c[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) c[m] = (aw[2 * m] & ((1u << MT) - 1u)) | ((aw[2 * m + 1] & ((1u << MT) - 1u)) << 16);
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            u32x4 A1;
#pragma unroll
            for (int m = 0; m < 4; ++m) A1[m] = __umul24((c[m] >> t) & 0x00010001u, 0x3F80u);         // bf16 1.0 / 0.0 pairs of rows 2 m, 2 m + 1
            const bf16x8 a1 = __builtin_bit_cast(bf16x8, A1);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, o.bq[u][2], acc[t][u], 0, 0, 0);      // smallest terms first
                acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, o.bq[u][1], acc[t][u], 0, 0, 0);
                acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, o.bq[u][0], acc[t][u], 0, 0, 0);
            }
        }
        if (want_cs) {
#pragma unroll
            for (int jj = 0; jj < rpg; ++jj) {
                const uint32_t wv = o.cw[jj] >> wsh;
#pragma unroll
                for (int t = 0; t < MT; ++t) cs[t] += (wv >> t) & 1u ? o.cd[jj] : 0.f;
            }
        }
    };
    {
        // row stages AND endpoint stages ping-pong over a loop unrolled by two (nsteps is even): every loop-carried register has ONE
        // definition inside the loop, so nothing is copied across the back edge -- a copy of a load's result is a wait for it
        S1 ia = load_idx(0);
        S2 ra = load_rows(ia);
        S1 ib = load_idx(1);
        S2 rb = load_rows(ib);
        ia = load_idx(2);
        publish(ra, 0, st0);
        int sa = 0, sb = 1;                                  // stage of step i, of step i + 1
        // The barrier keeps a K-group's waves in step, and a SIMD holds one wave of each group: with one barrier per step both would
        // multiply at the same time (sharing the matrix pipe) and then both publish (the pipe idle) -- measured 1.5 us per step.  Two
        // barriers per step, and group 1 runs HALF A STEP behind (one barrier more before its loop, one fewer after): while a SIMD's
        // group-0 wave multiplies, its group-1 wave loads, splits and publishes, and vice versa.
        if (kg == 1) __syncthreads();
        Ops op;
#pragma unroll 1
        for (int i = 0; i < nsteps; i += 2) {
            ra = load_rows(ia);                              // rows of step i + 2
            ib = load_idx(i + 3);                            // endpoints, dz, mask word of step i + 3
            publish(rb, i + 1, st0 + sb * kStage);           // step i + 1
            __syncthreads();
            fetch(st0 + sa * kStage, op);                    // step i
            mma(op);
            __syncthreads();
            sa = sb; sb = sb == 2 ? 0 : sb + 1;
            rb = load_rows(ib);                              // rows of step i + 3
            ia = load_idx(i + 4);
            publish(ra, i + 2, st0 + sb * kStage);           // step i + 2
            __syncthreads();
            fetch(st0 + sa * kStage, op);
            mma(op);
            __syncthreads();
            sa = sb; sb = sb == 2 ? 0 : sb + 1;
        }
        if (kg == 0) __syncthreads();
    }
    // ---- sum of the two K-groups (fixed order: group 0 + group 1), one tile row at a time through the (now dead) stages
    if (want_cs) {
#pragma unroll
        for (int t = 0; t < MT; ++t) cs[t] += __shfl_xor(cs[t], 32, 64);
    }
    dzs += __shfl_xor(dzs, 32, 64);
    float* red = reinterpret_cast<float*>(lds);
    constexpr int kRedCs = 4 * 2 * 16 * 64, kRedDz = kRedCs + 4 * MT * 64;
    static_assert(kRedDz + 8 <= 2 * 3 * kStage, "the K-group sum reuses the stages");
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        __syncthreads();                                     // t = 0: every wave is through its last multiply; t = 1: group 0 has read tile row 0
        if (t == 0 && lane == 0) red[kRedDz + wave] = dzs;
        if (kg == 1) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[((w * 2 + u) * 16 + r) * 64 + lane] = acc[t][u][r];
            if (t == 0) {
#pragma unroll
                for (int tt = 0; tt < MT; ++tt) red[kRedCs + (w * MT + tt) * 64 + lane] = cs[tt];
            }
        }
        __syncthreads();
        if (kg == 0) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][u][r] += red[((w * 2 + u) * 16 + r) * 64 + lane];
            if (t == 0) {
#pragma unroll
                for (int tt = 0; tt < MT; ++tt) cs[tt] += red[kRedCs + (w * MT + tt) * 64 + lane];
                if (want_dz && wave == 0 && lane == 0) {
                    float a = 0.f;
                    for (int z = 0; z < 8; ++z) a += red[kRedDz + z];
                    dzpart[slab_id] = a;
                }
            }
        }
    }
    if (kg != 0) return;
    float* out = slab + static_cast<int64_t>(slab_id) * M * N;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int h = 32 * MT * w + MT * ((r & 3) + 8 * (r >> 2) + 4 * g) + t;
            *reinterpret_cast<float2*>(out + static_cast<int64_t>(h) * N + n0 + 2 * l31) = make_float2(acc[t][0][r], acc[t][1][r]);
        }
    if (want_cs && g == 0) {
        float* cp = cpart + (static_cast<int64_t>(slab_id) * NG + ng) * M + 32 * MT * w + MT * l31;
#pragma unroll
        for (int t = 0; t < MT; ++t) cp[t] = cs[t];
    }
}

constexpr int kMaskfeatMaxSlabs = 128;
inline bool use_tall(int64_t K, int64_t M, int64_t N) { return K >= kTallK && M % 4 == 0 && N % 2 == 0; }
inline int pick_ksplit_tall(int64_t K, int64_t M, int64_t N) {
    const int64_t tiles = cdiv(M, 128) * cdiv(N, 64);
    int ks = 1;
    while (ks < 512 && tiles * ks < 1024 && K / (ks * 2) >= 256) ks *= 2;     // one wave per SIMD (two are slower: twice the slab traffic), >= 256 rows per slice
    return ks;
}

inline int pick_ksplit(int64_t K, int64_t M, int64_t N) {
    if (use_tall(K, M, N)) return pick_ksplit_tall(K, M, N);
    const int64_t tiles = cdiv(M, 32) * cdiv(N, 32);
    int ks = 1;
    while (ks < 64 && tiles * ks < 1024 && K / (ks * 2) >= 64) ks *= 2;      // fill ~1024 SIMDs, keep >= 64 rows per slice
    return ks;
}

}  // namespace
}  // namespace sgs

using namespace sgs;

extern "C" {

size_t sgs_gemm_tn_workspace_bytes(int64_t K, int64_t M, int64_t N) {
    if (K < 0 || M < 0 || N < 0) return 256;
    const size_t ks = static_cast<size_t>(pick_ksplit(K, M, N));
    return carve_bytes(ks * M * N, 4) + carve_bytes(ks * M, 4) + carve_bytes(kMaskfeatMaxSlabs * 8 * M + kMaskfeatMaxSlabs, 4) + 512;
    // split-K slabs + column-sum partials (+ the shared-operand kernel's: up to 8 column groups per slab, and its dz sums)
}

static int gemm_tn_impl(const float* A, const float* B, int64_t K, int64_t M, int64_t N, float* C, float* colsum_A, void* ws,
                        size_t ws_bytes, hipStream_t stream, int64_t ldc = 0, const uint32_t* Abits = nullptr, const float* dz = nullptr,
                        const float* rowscale = nullptr, float scale = 1.f, float* dz_sum = nullptr, float* C_raw = nullptr,
                        float* colsum_raw = nullptr, const int32_t* sd = nullptr);
// tall-K shapes: 1 = bf16x6 kernel (default; measured 147 vs 175 us incl. the 34 us slab reduction at K = 100 000, M = N = 256: the
// operand splits, 264 vector instructions per 48 MFMAs, are at the budget the matrix pipe leaves), 0 = fp32-MFMA kernel
static int g_tall_bf16x6 = 1;
// sgs_gemm_tn_mask_gather: 1 = the shared-operand kernel (gemm_tn_maskfeat, default), 0 = the per-wave-slice kernel; slabs: 0 = automatic
static int g_gather_shared = 1, g_gather_slabs = 0;
void sgs_gemm_tn_set_gather_variant(int shared, int slabs) { g_gather_shared = shared ? 1 : 0; g_gather_slabs = slabs > 0 ? slabs : 0; }
void sgs_gemm_tn_set_tall_variant(int v) { g_tall_bf16x6 = v < 0 ? 1 : (v ? 1 : 0); }

int sgs_gemm_tn(const float* A, const float* B, int64_t K, int64_t M, int64_t N, float* C, void* ws, size_t ws_bytes,
                sgs_stream_t stream_) {
    return gemm_tn_impl(A, B, K, M, N, C, nullptr, ws, ws_bytes, static_cast<hipStream_t>(stream_));
}

int sgs_gemm_tn_can_colsum(int64_t K, int64_t M, int64_t N) { return use_tall(K, M, N) && pick_ksplit_tall(K, M, N) > 1 ? 1 : 0; }

int sgs_gemm_tn_colsum(const float* A, const float* B, int64_t K, int64_t M, int64_t N, float* C, float* colsum_A, void* ws,
                       size_t ws_bytes, sgs_stream_t stream_) {
    SGS_REQUIRE(colsum_A && sgs_gemm_tn_can_colsum(K, M, N), SGS_EINVAL,
                "sgs_gemm_tn_colsum: shape not served by the tall-K kernel (check sgs_gemm_tn_can_colsum)");
    return gemm_tn_impl(A, B, K, M, N, C, colsum_A, ws, ws_bytes, static_cast<hipStream_t>(stream_));
}

int sgs_gemm_tn_ld(const float* A, const float* B, int64_t K, int64_t M, int64_t N, float* C, int64_t ldc, float* colsum_A, void* ws,
                   size_t ws_bytes, sgs_stream_t stream_) {
    SGS_REQUIRE(ldc >= N, SGS_EINVAL, "sgs_gemm_tn_ld: ldc < N");
    SGS_REQUIRE(!colsum_A || sgs_gemm_tn_can_colsum(K, M, N), SGS_EINVAL, "sgs_gemm_tn_ld: column sums only for the tall-K shapes");
    return gemm_tn_impl(A, B, K, M, N, C, colsum_A, ws, ws_bytes, static_cast<hipStream_t>(stream_), ldc);
}

/* C[M, N] (row stride ldc) = (diag(dz) mask diag(rowscale * scale))^T B with the mask given as bits [K, M/32] (bit m of row k): the weight
 * gradient d W1a = dv^T feat of the scorer with dv in its mask form (sgs_edge_score_bwd_core_bits); colsum_A (optional) = the column sums of
 * that dv = d b1.  Served by the tall-K bf16 kernel only: sgs_gemm_tn_mask_supported. */
int sgs_gemm_tn_mask_supported(int64_t K, int64_t M, int64_t N) {
    if (!use_tall(K, M, N) || M % 128 != 0 || N % 64 != 0) return 0;      // whole 128 x 64 wave tiles: the operands are read unconditionally
    const int ks = pick_ksplit_tall(K, M, N);
    return (ks >= 4 && ks % 4 == 0) ? 1 : 0;
}

int sgs_gemm_tn_mask(const uint32_t* Abits, const float* dz, const float* rowscale, float scale, const float* B, int64_t K, int64_t M, int64_t N,
                     float* C, int64_t ldc, float* colsum_A, float* dz_sum, float* C_raw, float* colsum_raw, void* ws, size_t ws_bytes,
                     sgs_stream_t stream_) {
    SGS_REQUIRE(ldc >= N, SGS_EINVAL, "sgs_gemm_tn_mask: ldc < N");
    SGS_REQUIRE(Abits && dz && rowscale, SGS_EINVAL, "sgs_gemm_tn_mask: null pointer");
    SGS_REQUIRE(sgs_gemm_tn_mask_supported(K, M, N), SGS_EINVAL, "sgs_gemm_tn_mask: shape not served (check sgs_gemm_tn_mask_supported)");
    SGS_REQUIRE(!colsum_raw || colsum_A, SGS_EINVAL, "sgs_gemm_tn_mask: colsum_raw needs colsum_A");
    return gemm_tn_impl(nullptr, B, K, M, N, C, colsum_A, ws, ws_bytes, static_cast<hipStream_t>(stream_), ldc, Abits, dz, rowscale, scale, dz_sum,
                        C_raw, colsum_raw);
}

/* The same product with B = codes[src] * codes[dst] gathered per row (sd [K, 2] int32 endpoints; `codes` [*, N]): the scorer's weight
 * gradient without a materialised feat.  Bit-identical to sgs_gemm_tn_mask on the materialised rows. */
int sgs_gemm_tn_mask_gather(const uint32_t* Abits, const float* dz, const float* rowscale, float scale, const float* codes, int64_t codes_rows,
                            const int32_t* sd, int64_t K, int64_t M, int64_t N, float* C, int64_t ldc, float* colsum_A, float* dz_sum, float* C_raw,
                            float* colsum_raw, void* ws, size_t ws_bytes, sgs_stream_t stream_) {
    SGS_REQUIRE(ldc >= N, SGS_EINVAL, "sgs_gemm_tn_mask_gather: ldc < N");
    SGS_REQUIRE(codes_rows > 0 && codes_rows * N < (int64_t(1) << 32), SGS_EINVAL, "sgs_gemm_tn_mask_gather: the codes table needs 32-bit element offsets");
    SGS_REQUIRE(Abits && dz && rowscale && codes && sd, SGS_EINVAL, "sgs_gemm_tn_mask_gather: null pointer");
    SGS_REQUIRE(sgs_gemm_tn_mask_supported(K, M, N), SGS_EINVAL, "sgs_gemm_tn_mask_gather: shape not served (check sgs_gemm_tn_mask_supported)");
    SGS_REQUIRE(!colsum_raw || colsum_A, SGS_EINVAL, "sgs_gemm_tn_mask_gather: colsum_raw needs colsum_A");
    return gemm_tn_impl(nullptr, codes, K, M, N, C, colsum_A, ws, ws_bytes, static_cast<hipStream_t>(stream_), ldc, Abits, dz, rowscale, scale,
                        dz_sum, C_raw, colsum_raw, sd);
}

static int gemm_tn_impl(const float* A, const float* B, int64_t K, int64_t M, int64_t N, float* C, float* colsum_A, void* ws,
                        size_t ws_bytes, hipStream_t stream, int64_t ldc, const uint32_t* Abits, const float* dz, const float* rowscale,
                        float scale, float* dz_sum, float* C_raw, float* colsum_raw, const int32_t* sd) {
    SGS_REQUIRE(K >= 0 && M >= 0 && N >= 0 && M < (1 << 30) && N < (1 << 30), SGS_EINVAL, "sgs_gemm_tn: bad sizes");
    if (M == 0 || N == 0) return SGS_OK;
    SGS_REQUIRE(C && (K == 0 || ((A || Abits) && B)), SGS_EINVAL, "sgs_gemm_tn: null pointer");
    SGS_REQUIRE(ws && ws_bytes >= sgs_gemm_tn_workspace_bytes(K, M, N), SGS_EWORKSPACE, "sgs_gemm_tn: workspace too small");
    const int ks = pick_ksplit(K, M, N);
    Carver cv(ws);
    float* slab = cv.take<float>(static_cast<size_t>(ks) * M * N);
    float* cpart = cv.take<float>(static_cast<size_t>(ks) * M);
    const bool strided = ldc > 0 && ldc != N;                  // the tile kernels write dense [M, N]: a strided C goes through the reduce
    float* dst = (ks == 1 && !strided) ? C : slab;
    int n_slabs = ks;                                          // slabs the reduction launch sums (= K-slices unless waves share a workgroup)
    if (Abits) {
        constexpr int NW = 4;
        constexpr size_t lds = (NW / 2) * (128 * 64 + 4 * 64) * sizeof(float);
        static bool raised_m = false;
        if (!raised_m) {
            SGS_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_tall_bf16x6<NW, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           static_cast<int>(lds)));
            raised_m = true;
        }
        n_slabs = ks / NW;
        // cpart holds ks * M floats and the NW-wave workgroups fill n_slabs * M of them: the K-slices' dz sums go behind those
        float* dzpart = dz_sum ? cpart + static_cast<size_t>(n_slabs) * M : nullptr;
        const int NGc = static_cast<int>(N / 64);
        if (sd && g_gather_shared && (M == 128 || M == 256) && N % 64 == 0 && (NGc == 1 || NGc == 2 || NGc == 4 || NGc == 8) && K >= 64) {
            // shared-operand kernel: one slab per 8-wave workgroup, NGc column-sum rows per slab
            int ns = g_gather_slabs > 0 ? g_gather_slabs : 256 / NGc;                   // one workgroup (two waves per SIMD) per CU
            while (ns > 1 && K / ns < 64) ns >>= 1;                                      // >= two steps per K-group
            if (ns > kMaskfeatMaxSlabs) ns = kMaskfeatMaxSlabs;
            if (ns > ks) ns = ks;                                                        // the slab carving is sized for ks slices
            float* cpart2 = cv.take<float>(static_cast<size_t>(kMaskfeatMaxSlabs) * 8 * M + kMaskfeatMaxSlabs);
            float* dzp = dz_sum ? cpart2 + static_cast<size_t>(ns) * NGc * M : nullptr;
            const size_t lds_b = static_cast<size_t>(2 * 3 * (2 * 3 * 64 * 4 + (M / 32) * 16 + 16)) * 4;
            const dim3 grid(static_cast<unsigned>(NGc), static_cast<unsigned>(ns));
            float* cpa = (colsum_A || dz_sum) ? cpart2 : static_cast<float*>(nullptr);
            const int2* sd2 = reinterpret_cast<const int2*>(sd);
            const int Ni = static_cast<int>(N);
#define SGS_MASKFEAT(MT_, NG_) hipLaunchKernelGGL((gemm_tn_maskfeat<MT_, NG_>), grid, dim3(512), lds_b, stream, Abits, dz, B, sd2, K, Ni, slab, cpa, dzp)
            if (M == 256) {
                if (NGc == 1) SGS_MASKFEAT(2, 1); else if (NGc == 2) SGS_MASKFEAT(2, 2); else if (NGc == 4) SGS_MASKFEAT(2, 4); else SGS_MASKFEAT(2, 8);
            } else {
                if (NGc == 1) SGS_MASKFEAT(1, 1); else if (NGc == 2) SGS_MASKFEAT(1, 2); else if (NGc == 4) SGS_MASKFEAT(1, 4); else SGS_MASKFEAT(1, 8);
            }
#undef SGS_MASKFEAT
            hipLaunchKernelGGL(gemm_tn_reduce, dim3(cdiv(M * N + M + 1, 256)), dim3(256), 0, stream, slab, M * N, ns, C,
                               static_cast<const float*>(colsum_A ? cpart2 : nullptr), static_cast<int>(M), colsum_A, static_cast<int>(N),
                               ldc > 0 ? ldc : N, rowscale, scale, static_cast<const float*>(dzp), ns, dz_sum, C_raw, colsum_raw, ns * NGc);
            SGS_LAUNCH_OK();
            return SGS_OK;
        }
        if (sd) {
            static bool raised_g = false;
            if (!raised_g) {
                SGS_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_tall_bf16x6<NW, true, true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
                raised_g = true;
            }
            hipLaunchKernelGGL((gemm_tn_tall_bf16x6<NW, true, true>), dim3(cdiv(M, 128), cdiv(N, 64), n_slabs), dim3(64 * NW), lds, stream, A, B, K,
                               static_cast<int>(M), static_cast<int>(N), ks, slab, (colsum_A || dz_sum) ? cpart : static_cast<float*>(nullptr), Abits,
                               dz, dzpart, sd);
        } else
        hipLaunchKernelGGL((gemm_tn_tall_bf16x6<NW, true>), dim3(cdiv(M, 128), cdiv(N, 64), n_slabs), dim3(64 * NW), lds, stream, A, B, K,
                           static_cast<int>(M), static_cast<int>(N), ks, slab, (colsum_A || dz_sum) ? cpart : static_cast<float*>(nullptr), Abits, dz,
                           dzpart);
        hipLaunchKernelGGL(gemm_tn_reduce, dim3(cdiv(M * N + M + 1, 256)), dim3(256), 0, stream, slab, M * N, n_slabs, C,
                           static_cast<const float*>(colsum_A ? cpart : nullptr), static_cast<int>(M), colsum_A, static_cast<int>(N),
                           ldc > 0 ? ldc : N, rowscale, scale, static_cast<const float*>(dzpart), ks, dz_sum, C_raw, colsum_raw);
        SGS_LAUNCH_OK();
        return SGS_OK;
    }
    if (use_tall(K, M, N) && g_tall_bf16x6 && ks >= 4 && ks % 4 == 0) {
        constexpr int NW = 4;
        constexpr size_t lds = (NW / 2) * (128 * 64 + 4 * 64) * sizeof(float);
        static bool raised = false;
        if (!raised) {
            SGS_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_tall_bf16x6<NW>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           static_cast<int>(lds)));
            raised = true;
        }
        n_slabs = ks / NW;
        hipLaunchKernelGGL((gemm_tn_tall_bf16x6<NW>), dim3(cdiv(M, 128), cdiv(N, 64), n_slabs), dim3(64 * NW), lds, stream, A, B, K,
                           static_cast<int>(M), static_cast<int>(N), ks, slab, colsum_A ? cpart : static_cast<float*>(nullptr));
    } else if (use_tall(K, M, N) && g_tall_bf16x6)
        hipLaunchKernelGGL((gemm_tn_tall_bf16x6<1>), dim3(cdiv(M, 128), cdiv(N, 64), ks), dim3(64), 0, stream, A, B, K, static_cast<int>(M),
                           static_cast<int>(N), ks, dst, colsum_A ? cpart : static_cast<float*>(nullptr));
    else if (use_tall(K, M, N))
        hipLaunchKernelGGL(gemm_tn_tall_tile, dim3(cdiv(M, 128), cdiv(N, 64), ks), dim3(64), 0, stream, A, B, K, static_cast<int>(M),
                           static_cast<int>(N), ks, dst, colsum_A ? cpart : static_cast<float*>(nullptr));
    else if (ks == 2 || ks == 4 || ks == 8) {
        // partition-sized K: the K-slices are the waves of one workgroup, reduced in LDS, C written in place (any ldc)
        const dim3 grid(cdiv(M, 32), cdiv(N, 32));
        const int64_t ld = ldc > 0 ? ldc : N;
        if (ks == 8 && K >= 512)      // sixteen 64-row slices: half the dependent load batches per wave (the launch is latency-bound)
            hipLaunchKernelGGL((gemm_tn_wg<16>), grid, dim3(1024), 0, stream, A, B, K, static_cast<int>(M), static_cast<int>(N), C, ld);
        else if (ks == 8) hipLaunchKernelGGL((gemm_tn_wg<8>), grid, dim3(512), 0, stream, A, B, K, static_cast<int>(M), static_cast<int>(N), C, ld);
        else if (ks == 4) hipLaunchKernelGGL((gemm_tn_wg<4>), grid, dim3(256), 0, stream, A, B, K, static_cast<int>(M), static_cast<int>(N), C, ld);
        else              hipLaunchKernelGGL((gemm_tn_wg<2>), grid, dim3(128), 0, stream, A, B, K, static_cast<int>(M), static_cast<int>(N), C, ld);
        SGS_LAUNCH_OK();
        return SGS_OK;
    } else
        hipLaunchKernelGGL(gemm_tn_tile, dim3(cdiv(M, 32), cdiv(N, 32), ks), dim3(64), 0, stream, A, B, K, static_cast<int>(M),
                           static_cast<int>(N), ks, dst);
    if (ks > 1 || strided)
        hipLaunchKernelGGL(gemm_tn_reduce, dim3(cdiv(M * N + (colsum_A ? M : 0), 256)), dim3(256), 0, stream, slab, M * N, n_slabs, C,
                           static_cast<const float*>(colsum_A ? cpart : nullptr), static_cast<int>(M), colsum_A, static_cast<int>(N), ldc);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

}  // extern "C"
