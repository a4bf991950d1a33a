// Shared host/device helpers for libsgs_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/sgs_hip.h"

namespace sgs {

constexpr int kWave = 64;

// ---------------------------------------------------------------- errors (thread-local message)
void set_error(const char* fmt, ...);

#define SGS_REQUIRE(cond, code, ...)      \
    do {                                  \
        if (!(cond)) {                    \
            ::sgs::set_error(__VA_ARGS__); \
            return (code);                \
        }                                 \
    } while (0)

#define SGS_HIP_OK(expr)                                                                  \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            ::sgs::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return SGS_EHIP;                                                              \
        }                                                                                 \
    } while (0)

#define SGS_LAUNCH_OK()                                                                   \
    do {                                                                                  \
        hipError_t _e = hipGetLastError();                                                \
        if (_e != hipSuccess) {                                                           \
            ::sgs::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
            return SGS_EHIP;                                                              \
        }                                                                                 \
    } while (0)

// ---------------------------------------------------------------- RNG epoch (HIP-graph replay)
// Seeds are kernel arguments by value, which a captured graph freezes.  When the host registers a device word
// (sgs_rng_set_epoch_buffer), every RNG-consuming kernel folds *epoch into its seed, so a captured `epoch += 1`
// makes each replay draw fresh noise / dropout masks.  nullptr (default) = seeds used as given.
const uint64_t* epoch_ptr();
__device__ __forceinline__ uint64_t fold_epoch(uint64_t seed, const uint64_t* __restrict__ epoch) {
    return epoch ? seed + 0x9E3779B97F4A7C15ULL * (*epoch) : seed;
}

// ---------------------------------------------------------------- dynamic candidate-edge count (HIP-graph replay)
// Sizes are kernel arguments by value, which a captured graph freezes.  While a device word is registered
// (sgs_dyn_edges_set), sgs_sample_topq and sgs_edge_score_fwd read the number of candidate edges E from it at run time and
// treat their `E` argument as the capacity (grid size, row stride of edge_index, buffer sizes): ONE captured step then
// serves partitions of any size.  nullptr (default) = sizes used as given.
const int64_t* dyn_edges_ptr();

// ---------------------------------------------------------------- zero fill as a KERNEL
// hipMemsetAsync is deliberately not used anywhere in this library: captured into a HIP graph it becomes a memset
// node, and on replay (ROCm 7.2 / gfx950) those were observed not to stay ordered with the neighbouring kernel nodes
// (the sampler's select state was cleared after the kernels that fill it).  A kernel node has no such problem and
// several adjacent scratch words are cleared by one launch.  `p` 4-byte aligned, `bytes` a multiple of 4.
int zero_async(void* p, size_t bytes, hipStream_t stream);
// two spans in one launch: 32-bit word `v1` over [p1, p1+bytes1), `v2` over [p2, p2+bytes2)
int fill2_async(void* p1, size_t bytes1, uint32_t v1, void* p2, size_t bytes2, uint32_t v2, hipStream_t stream);

// ---------------------------------------------------------------- workspace carving (256-B aligned)
struct Carver {
    char* base;
    size_t off;
    explicit Carver(void* p) : base(static_cast<char*>(p)), off(0) {}
    template <typename T>
    T* take(size_t n) {
        size_t bytes = (n * sizeof(T) + 255) & ~size_t(255);
        T* r = reinterpret_cast<T*>(base + off);
        off += bytes;
        return r;
    }
};
inline size_t carve_bytes(size_t n, size_t elem) { return (n * elem + 255) & ~size_t(255); }

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------- device: wave / block reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;  // valid in lane 0
}
__device__ __forceinline__ float wave_sum_all(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max_all(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_sum_int_all(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// The same reductions on the DPP path (no LDS crossbar traffic; __shfl_xor compiles to ds_bpermute_b32 + a wait on this target, which
// dominated kernels that reduce per edge).  Used where the summation ORDER is free (gradient kernels); the sampler's normalisers keep
// wave_sum above so that their bits do not move.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_get(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
// every lane ends with the sum over its 16-lane row (quad_perm xor 1, xor 2, row_half_mirror, row_mirror)
__device__ __forceinline__ float row16_sum_all_dpp(float v) {
    v += dpp_get<0xB1>(v);
    v += dpp_get<0x4E>(v);
    v += dpp_get<0x141>(v);
    v += dpp_get<0x140>(v);
    return v;
}
// sum over each 32-lane half of the wave, valid in lanes 16..31 / 48..63 (row_bcast15 into rows 1 and 3)
__device__ __forceinline__ float half_wave_sum_hi(float v) {
    v = row16_sum_all_dpp(v);
    v += dpp_get<0x142, 0xA>(v);
    return v;
}
// sum over the wave, valid in lane 63 (row_bcast31 into rows 2 and 3 on top)
__device__ __forceinline__ float wave_sum_hi_dpp(float v) {
    v = half_wave_sum_hi(v);
    v += dpp_get<0x143, 0xC>(v);
    return v;
}

// Deterministic block sum (fixed tree): result valid in thread 0.  `red` needs blockDim/64 floats.
__device__ __forceinline__ float block_sum(float v, float* red) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0)
        for (int i = 0; i < nw; ++i) r += red[i];
    __syncthreads();
    return r;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_max_all(v);
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float r = red[0];
    for (int i = 1; i < nw; ++i) r = fmaxf(r, red[i]);
    __syncthreads();
    return r;  // valid in every thread
}

// ---------------------------------------------------------------- fp32 on the bf16 matrix pipe (bf16x6)
// v = v1 + v2 + v3 exactly, three bf16 pieces by round-to-nearest residuals; users: csrc/edge_score.hip (the scheme and its
// error bound are described there), csrc/gemm_tn.hip.
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {       // (bf16(a) low, bf16(b) high), round to nearest even
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
// (a, b) -> packed pieces p1, p2, p3 with a = a1 + a2 + a3 (same for b)
__device__ __forceinline__ void split3(float a, float b, uint32_t& p1, uint32_t& p2, uint32_t& p3) {
    p1 = pk_bf16(a, b);
    const float ra = a - __uint_as_float(p1 << 16), rb = b - __uint_as_float(p1 & 0xFFFF0000u);
    p2 = pk_bf16(ra, rb);
    const float sa = ra - __uint_as_float(p2 << 16), sb = rb - __uint_as_float(p2 & 0xFFFF0000u);
    p3 = pk_bf16(sa, sb);
}


// ---------------------------------------------------------------- counter-based RNG
// Philox4x32-10 (Salmon et al. 2011).  Integer-only, so host restatements reproduce it exactly.
struct Philox4 {
    uint32_t v[4];
};
__host__ __device__ __forceinline__ uint32_t mulhi32(uint32_t a, uint32_t b) {
    return static_cast<uint32_t>((static_cast<uint64_t>(a) * b) >> 32);
}
__host__ __device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                          uint32_t k0, uint32_t k1) {
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = mulhi32(M0, c0), lo0 = M0 * c0;
        uint32_t hi1 = mulhi32(M1, c2), lo1 = M1 * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    return Philox4{{c0, c1, c2, c3}};
}

// Exp(1) draw number `e` of stream (seed, stream_id): noise = -log1p(-v) with v = (x + 1/2) 2^-32 from all 32 random bits.
// The race's WINNERS are the edges with the smallest noise, so that tail must not be coarse: v keeps 24 significant bits down to
// 2^-33 (noise resolved to ~1e-17 near 0; a 23-bit uniform would quantise it to 1.2e-7 and bias draws with q / E below ~1e-4).
// v is clamped below 1 (noise <= 16.6: a tail of probability 6e-8 that can only lose the race); noise is never 0.
__device__ __forceinline__ float exp_noise_at(uint64_t seed, uint64_t stream_id, uint64_t e) {
    const uint64_t blk = e >> 2;
    Philox4 r = philox4x32_10(static_cast<uint32_t>(blk), static_cast<uint32_t>(blk >> 32),
                              static_cast<uint32_t>(stream_id), static_cast<uint32_t>(stream_id >> 32),
                              static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32));
    const uint32_t x = r.v[e & 3];
    const float v = fminf((static_cast<float>(x) + 0.5f) * 2.3283064365386963e-10f, 0.99999994f);
    return -log1pf(-v);
}

// Dropout keep decision for element (row, col) of dropout site `site`: a 64-bit murmur-style mix of
// (seed, site, row) gives a 32-bit row key (computed once per row), then a 32-bit murmur3 finaliser
// of (row key, column quad) gives 16-bit draws compared against p * 2^16 (dropout_pair_bits below).  Every fused kernel and the materialising
// sgs_dropout_keep use these two functions, so recompute-in-backward sees the same bits as forward.
__host__ __device__ __forceinline__ uint32_t mix64to32(uint64_t z) {
    z ^= z >> 33; z *= 0xff51afd7ed558ccdULL;
    z ^= z >> 33; z *= 0xc4ceb9fe1a85ec53ULL;
    z ^= z >> 33;
    return static_cast<uint32_t>(z >> 32);
}
__host__ __device__ __forceinline__ uint32_t dropout_row_key(uint64_t seed, uint32_t site, uint64_t row) {
    uint64_t z = seed ^ (0x9E3779B97F4A7C15ULL * (static_cast<uint64_t>(site) + 1));
    z = (z ^ row) * 0xD6E8FEB86659FD93ULL;
    return mix64to32(z);
}
// One murmur3 finaliser per FOUR columns (round 3; per two before): its 32 bits are the 16-bit draws of columns 4 j and 4 j + 1, and one
// more multiply-xorshift of it gives those of columns 4 j + 2 and 4 j + 3.  The scorer's epilogue spends ~45 % of its vector issue slots on
// these hashes (two quarter-rate v_mul_lo_u32 each); sharing one across a quad cuts that by 30 %.  Checked on 120 000 rows x 256 columns
// at p = 0.3: keep rates 0.697-0.704, column-column correlation of the keep bits max 0.012 / rms 0.0029 (= 1 / sqrt(rows), the same as with
// a finaliser per pair; single-multiply mixers gave 0.07-0.30 and were rejected).  P(drop) = round(p * 65536) / 65536.
// Callers ask per column pair (columns 2 j', 2 j' + 1): the two pairs of a quad inline to ONE finaliser.
__host__ __device__ __forceinline__ uint32_t dropout_quad_hash(uint32_t row_key, uint32_t col_quad) {
    uint32_t h = row_key ^ (col_quad * 0x9E3779B1u);
    h ^= h >> 16; h *= 0x85EBCA6Bu;
    h ^= h >> 13; h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}
__host__ __device__ __forceinline__ uint32_t dropout_pair_bits(uint32_t row_key, uint32_t col_pair) {
    const uint32_t h = dropout_quad_hash(row_key, col_pair >> 1);
    if ((col_pair & 1u) == 0u) return h;
    uint32_t g = h * 0x9E3779B1u;
    g ^= g >> 16;
    return g;
}
__host__ __device__ __forceinline__ bool dropout_keep_col(uint32_t row_key, uint32_t col, uint32_t thresh16 /* = p * 2^16 */) {
    const uint32_t h = dropout_pair_bits(row_key, col >> 1);
    return ((col & 1u) ? (h >> 16) : (h & 0xFFFFu)) >= thresh16;
}
__host__ __device__ __forceinline__ bool dropout_keep_at(uint64_t seed, uint32_t site, uint64_t row, uint32_t col,
                                                         uint32_t thresh) {
    return dropout_keep_col(dropout_row_key(seed, site, row), col, thresh);
}
__host__ __device__ __forceinline__ uint32_t dropout_thresh(float p) {
    double t = static_cast<double>(p) * 65536.0 + 0.5;
    if (t < 0) t = 0;
    if (t > 65535.0) t = 65535.0;
    return static_cast<uint32_t>(t);
}

}  // namespace sgs
