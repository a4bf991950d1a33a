// K0 / K2 / K3: exponential-race top-q edge sampler with stable compaction (gfx950).
//
// Reference: sampling.py:91-155 (`gumbel_softmax_sampling`), training_hybrid.py:46-48 (prior
// draw), :83/:86 (boolean-mask compaction).  torch.multinomial(s, q, replacement=False) is
// topk(s / Exp(1)); we compute the fp32 keys with the reference's expression order (IEEE
// div/mul/add, no contraction), find the q-th largest key EXACTLY with a 3-digit radix select
// over the key bit patterns (keys >= 0, so uint order == float order), break ties towards the
// lowest edge id, and emit mask + compacted columns in original edge order.
//
// HBM-bound: per candidate edge 8 B read (+4 B noise if supplied) + 4 B key write/3x4 B re-read
// (L2/Infinity-Cache resident at partition scale) + 1 B mask; per selected edge 16 B index
// read + 28 B written.  All counting is integer => run-to-run and rank-count invariant.
#include "sgs_common.h"

namespace sgs {
namespace {

constexpr int kThreads = 256;
constexpr int kItems = 8;                    // consecutive elements per thread
constexpr int kChunk = kThreads * kItems;    // 2048 elements per block
constexpr int kBins = 2048;                  // 11-bit digits
constexpr int kShift0 = 21, kShift1 = 10, kShift2 = 0;
constexpr uint32_t kMask1 = 0x7FFu, kMask2 = 0x3FFu;

struct SelectState {
    uint32_t prefix;   // key bits decided so far (high digits)
    uint32_t k_rem;    // how many keys still to take among those matching `prefix`
    uint32_t n_gt;     // (final) number of keys strictly greater than the threshold
    uint32_t pad;
};

// ---------------------------------------------------------------- reductions (deterministic tree)
// MODE 0: sum(p)   MODE 1: max(p)   MODE 2: sum(exp(p - max))
template <int MODE>
__global__ void __launch_bounds__(kThreads) reduce_partial(const float* __restrict__ p, int64_t E,
                                                          const float* __restrict__ scal, float* __restrict__ part) {
    __shared__ float red[kThreads / 64];
    const int64_t base = static_cast<int64_t>(blockIdx.x) * kChunk;
    float mx = 0.f;
    if (MODE == 2) mx = scal[1];
    float acc = (MODE == 1) ? -INFINITY : 0.f;
#pragma unroll
    for (int i = 0; i < kItems; ++i) {
        const int64_t e = base + static_cast<int64_t>(i) * kThreads + threadIdx.x;
        if (e < E) {
            const float v = p ? p[e] : 1.f;          // p == nullptr: uniform weights (sgs_sample_topq, random_edge_sampling)
            if (MODE == 0) acc += v;
            else if (MODE == 1) acc = fmaxf(acc, v);
            else acc += expf(v - mx);
        }
    }
    if (MODE == 1) {
        const float r = block_max(acc, red);
        if (threadIdx.x == 0) part[blockIdx.x] = r;
    } else {
        const float r = block_sum(acc, red);
        if (threadIdx.x == 0) part[blockIdx.x] = r;
    }
}

template <int MODE>
__global__ void __launch_bounds__(kThreads) reduce_final(const float* __restrict__ part, int64_t n,
                                                        float* __restrict__ scal) {
    __shared__ float red[kThreads / 64];
    float acc = (MODE == 1) ? -INFINITY : 0.f;
    for (int64_t i = threadIdx.x; i < n; i += kThreads) {
        const float v = part[i];
        if (MODE == 1) acc = fmaxf(acc, v);
        else acc += v;
    }
    if (MODE == 1) {
        const float r = block_max(acc, red);
        if (threadIdx.x == 0) scal[1] = r;
    } else {
        const float r = block_sum(acc, red);
        if (threadIdx.x == 0) scal[0] = r;
    }
}

// ---------------------------------------------------------------- key computation
// The reference's fp32 expression order, op by op (sampling.py:93-96):
//   samples = edge_probs / (edge_probs.sum() + eps)
//   samples = (1 - c) * samples + c * batch.prob          (python doubles -> fp32 scalars)
//   keys    = samples / Exp(1)
template <int MODE>
__device__ __forceinline__ float sample_prob(float pv, float Zeps, float mx, float priorv, bool has_prior,
                                             float one_minus_c, float c) {
    if (MODE == SGS_SAMPLE_LEARNED) {
        float s = __fdiv_rn(pv, Zeps);
        if (has_prior) s = __fadd_rn(__fmul_rn(one_minus_c, s), __fmul_rn(c, priorv));
        return s;
    } else {
        return __fdiv_rn(expf(pv - mx), Zeps);
    }
}

template <int MODE>
__global__ void __launch_bounds__(kThreads) keys_hist0(const float* __restrict__ p, const float* __restrict__ prior,
                                                      const float* __restrict__ noise, uint64_t seed,
                                                      uint64_t stream_id, const uint64_t* __restrict__ epoch, int64_t edge_offset, int64_t E, float one_minus_c, float c,
                                                      const float* __restrict__ scal, uint32_t* __restrict__ keys,
                                                      float* __restrict__ keys_out, uint32_t* __restrict__ hist) {
    __shared__ uint32_t lh[kBins];
    for (int i = threadIdx.x; i < kBins; i += kThreads) lh[i] = 0;
    __syncthreads();
    seed = fold_epoch(seed, epoch);
    const float Z = scal[0];
    const float Zeps = (MODE == SGS_SAMPLE_LEARNED) ? __fadd_rn(Z, 1e-12f) : Z;
    const float mx = (MODE == SGS_SAMPLE_PRIOR) ? scal[1] : 0.f;
    const bool has_prior = prior != nullptr;
    // grid-stride over the 2048-edge chunks: at full-graph scale (56 k chunks) a workgroup per chunk would flush 56 k
    // private histograms into the 2048 global bins; a capped grid flushes a few thousand
    const int64_t nchunk = (E + kChunk - 1) / kChunk;
    for (int64_t chunk = blockIdx.x; chunk < nchunk; chunk += gridDim.x) {
        const int64_t base = chunk * kChunk;
#pragma unroll
        for (int i = 0; i < kItems; ++i) {
            const int64_t e = base + static_cast<int64_t>(i) * kThreads + threadIdx.x;
            if (e < E) {
                const float s = sample_prob<MODE>(p ? p[e] : 1.f, Zeps, mx, has_prior ? prior[e] : 0.f, has_prior, one_minus_c, c);
                const float nz = noise ? noise[e] : exp_noise_at(seed, stream_id, static_cast<uint64_t>(edge_offset + e));
                const float key = __fdiv_rn(s, nz);
                const uint32_t bits = __float_as_uint(key);
                keys[e] = bits;
                if (keys_out) keys_out[e] = key;
                atomicAdd(&lh[bits >> kShift0], 1u);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kBins; i += kThreads) {
        const uint32_t v = lh[i];
        if (v) atomicAdd(&hist[i], v);
    }
}

// Histogram of the next digit among keys whose higher digits equal state->prefix.
__global__ void __launch_bounds__(kThreads) hist_next(const uint32_t* __restrict__ keys, int64_t E, int shift,
                                                     uint32_t digit_mask, int prev_shift,
                                                     const SelectState* __restrict__ st, uint32_t* __restrict__ hist) {
    __shared__ uint32_t lh[kBins];
    for (int i = threadIdx.x; i < kBins; i += kThreads) lh[i] = 0;
    __syncthreads();
    const uint32_t want = st->prefix >> prev_shift;
    const int64_t nchunk = (E + kChunk - 1) / kChunk;
    for (int64_t chunk = blockIdx.x; chunk < nchunk; chunk += gridDim.x) {
        const int64_t base = chunk * kChunk;
#pragma unroll
        for (int i = 0; i < kItems; ++i) {
            const int64_t e = base + static_cast<int64_t>(i) * kThreads + threadIdx.x;
            if (e < E) {
                const uint32_t bits = keys[e];
                if ((bits >> prev_shift) == want) atomicAdd(&lh[(bits >> shift) & digit_mask], 1u);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kBins; i += kThreads) {
        const uint32_t v = lh[i];
        if (v) atomicAdd(&hist[i], v);
    }
}

// One block: locate the digit bin holding the k_rem-th largest key of the current prefix.
__device__ __forceinline__ uint32_t load_coherent(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ void select_digit_body(uint32_t* __restrict__ hist, int shift, int first, uint32_t q, SelectState* __restrict__ st) {
    __shared__ uint32_t tsum[kThreads];
    __shared__ uint32_t found_bin, found_above;
    constexpr int per = kBins / kThreads;  // 8 bins per thread, descending order
    const uint32_t k = first ? q : st->k_rem;
    uint32_t loc[per];
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < per; ++j) {
        const int bin = kBins - 1 - (threadIdx.x * per + j);
        loc[j] = load_coherent(hist + bin);
        s += loc[j];
    }
    tsum[threadIdx.x] = s;
    __syncthreads();
    // exclusive prefix over threads (descending bins): Hillis-Steele in LDS
    for (int off = 1; off < kThreads; off <<= 1) {
        uint32_t v = (threadIdx.x >= off) ? tsum[threadIdx.x - off] : 0u;
        __syncthreads();
        tsum[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t before = tsum[threadIdx.x] - s;  // keys in strictly higher bins than this thread's range
    if (before < k && before + s >= k) {
        uint32_t run = before;
#pragma unroll
        for (int j = 0; j < per; ++j) {
            if (run < k && run + loc[j] >= k) {
                found_bin = kBins - 1 - (threadIdx.x * per + j);
                found_above = run;
            }
            run += loc[j];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t pre = first ? 0u : st->prefix;
        st->prefix = pre | (found_bin << shift);
        st->k_rem = k - found_above;
        if (shift == 0) st->n_gt = q - (k - found_above);
    }
    // clear the histogram for the next pass
    for (int i = threadIdx.x; i < kBins; i += kThreads) hist[i] = 0;
    __syncthreads();
}
__global__ void __launch_bounds__(kThreads) select_digit(uint32_t* __restrict__ hist, int shift, int first,
                                                        uint32_t q, SelectState* __restrict__ st) {
    select_digit_body(hist, shift, first, q, st);
}

// ---------------------------------------------------------------- counting + scan + compaction
__device__ __forceinline__ void load_keys8(const uint32_t* __restrict__ keys, int64_t e0, int64_t E, uint32_t (&k)[kItems]) {
    if (e0 + kItems <= E) {
        const uint4 a = *reinterpret_cast<const uint4*>(keys + e0);
        const uint4 b = *reinterpret_cast<const uint4*>(keys + e0 + 4);
        k[0] = a.x; k[1] = a.y; k[2] = a.z; k[3] = a.w;
        k[4] = b.x; k[5] = b.y; k[6] = b.z; k[7] = b.w;
    } else {
#pragma unroll
        for (int j = 0; j < kItems; ++j) k[j] = (e0 + j < E) ? keys[e0 + j] : 0u;
    }
}

__global__ void __launch_bounds__(kThreads) count_blocks(const uint32_t* __restrict__ keys, int64_t E,
                                                        const SelectState* __restrict__ st, uint2* __restrict__ cnt) {
    __shared__ int red[2 * (kThreads / 64)];
    const uint32_t T = st->prefix;
    const int64_t e0 = static_cast<int64_t>(blockIdx.x) * kChunk + static_cast<int64_t>(threadIdx.x) * kItems;
    uint32_t k[kItems];
    load_keys8(keys, e0, E, k);
    int gt = 0, eq = 0;
#pragma unroll
    for (int j = 0; j < kItems; ++j) {
        const bool in = e0 + j < E;
        gt += (in && k[j] > T);
        eq += (in && k[j] == T);
    }
    gt = wave_sum_int_all(gt);
    eq = wave_sum_int_all(eq);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) { red[2 * wid] = gt; red[2 * wid + 1] = eq; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int g = 0, q_ = 0;
        for (int w = 0; w < kThreads / 64; ++w) { g += red[2 * w]; q_ += red[2 * w + 1]; }
        cnt[blockIdx.x] = make_uint2(static_cast<uint32_t>(g), static_cast<uint32_t>(q_));
    }
}

// One block: exclusive scan of the per-block (gt, eq) counts, in place.
__device__ void scan_blocks_body(uint2* __restrict__ cnt, int64_t nblk) {
    __shared__ uint32_t sg[kThreads], se[kThreads];
    const int64_t per = (nblk + kThreads - 1) / kThreads;
    const int64_t lo = static_cast<int64_t>(threadIdx.x) * per;
    const int64_t hi = (lo + per < nblk) ? lo + per : nblk;
    uint32_t g = 0, e = 0;
    for (int64_t i = lo; i < hi; ++i) { g += cnt[i].x; e += cnt[i].y; }
    sg[threadIdx.x] = g; se[threadIdx.x] = e;
    __syncthreads();
    for (int off = 1; off < kThreads; off <<= 1) {
        uint32_t vg = 0, ve = 0;
        if (threadIdx.x >= off) { vg = sg[threadIdx.x - off]; ve = se[threadIdx.x - off]; }
        __syncthreads();
        sg[threadIdx.x] += vg; se[threadIdx.x] += ve;
        __syncthreads();
    }
    uint32_t rg = sg[threadIdx.x] - g, re = se[threadIdx.x] - e;
    for (int64_t i = lo; i < hi; ++i) {
        const uint2 c = cnt[i];
        cnt[i] = make_uint2(rg, re);
        rg += c.x; re += c.y;
    }
}
__global__ void __launch_bounds__(kThreads) scan_blocks(uint2* __restrict__ cnt, int64_t nblk) { scan_blocks_body(cnt, nblk); }

// blk_gt / blk_eq: number of keys > T / == T in all chunks before this workgroup's
__device__ __forceinline__ void compact_body(uint32_t blk_gt, uint32_t blk_eq, const uint32_t* __restrict__ keys, int64_t E, int64_t Ecap, int64_t q,
                                             int64_t ties_override, int64_t eid_offset, const SelectState* __restrict__ st,
                                             const float* __restrict__ p, const int64_t* __restrict__ edge_index,
                                             uint8_t* __restrict__ mask, int mask_aligned,
                                             int64_t* __restrict__ sampled_eid, int64_t* __restrict__ sei,
                                             float* __restrict__ sampled_p) {
    __shared__ uint32_t wg[kThreads / 64], we[kThreads / 64];
    const uint32_t T = st->prefix, k_rem = ties_override >= 0 ? static_cast<uint32_t>(ties_override) : st->k_rem;
    const int64_t e0 = static_cast<int64_t>(blockIdx.x) * kChunk + static_cast<int64_t>(threadIdx.x) * kItems;
    uint32_t k[kItems];
    load_keys8(keys, e0, E, k);
    uint32_t gt = 0, eq = 0;
#pragma unroll
    for (int j = 0; j < kItems; ++j) {
        const bool in = e0 + j < E;
        gt += (in && k[j] > T);
        eq += (in && k[j] == T);
    }
    // exclusive prefix of (gt, eq) over the block's threads: wave scan + wave totals
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint32_t ig = gt, ie = eq;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t vg = __shfl_up(ig, o, 64), ve = __shfl_up(ie, o, 64);
        if (lane >= o) { ig += vg; ie += ve; }
    }
    if (lane == 63) { wg[wid] = ig; we[wid] = ie; }
    __syncthreads();
    uint32_t bg = blk_gt, be = blk_eq;
    for (int w = 0; w < wid; ++w) { bg += wg[w]; be += we[w]; }
    bg += ig - gt;   // #gt with lower id (global)
    be += ie - eq;   // #eq with lower id (global)
    uint64_t mbits = 0;
    // The endpoint columns of this thread's 8 candidates are loaded UNCONDITIONALLY as four 16-byte vectors per row of
    // edge_index when the tile is whole: at the usual 20 % keep rate a predicated 8-byte gather touches nearly every cache
    // line anyway, and at full-graph scale (E = 114.6 M) the streaming form is what HBM delivers at speed.
    const bool whole = e0 + kItems <= E && sei != nullptr && ((reinterpret_cast<uintptr_t>(edge_index) | (static_cast<uint64_t>(Ecap) * 8)) & 15) == 0;
    int64_t sv[kItems], dv[kItems];
    if (whole) {
#pragma unroll
        for (int j = 0; j < kItems; j += 2) {
            const longlong2 a = *reinterpret_cast<const longlong2*>(edge_index + e0 + j);
            const longlong2 b = *reinterpret_cast<const longlong2*>(edge_index + Ecap + e0 + j);
            sv[j] = a.x; sv[j + 1] = a.y; dv[j] = b.x; dv[j + 1] = b.y;
        }
    }
#pragma unroll
    for (int j = 0; j < kItems; ++j) {
        const int64_t e = e0 + j;
        if (e < E) {
            const bool isgt = k[j] > T, iseq = k[j] == T;
            const bool sel = isgt || (iseq && be < k_rem);
            if (sel) {
                mbits |= (uint64_t(1) << (8 * j));
                const int64_t pos = static_cast<int64_t>(bg) + static_cast<int64_t>(be < k_rem ? be : k_rem);
                if (sampled_eid) sampled_eid[pos] = eid_offset + e;
                if (sei) {
                    sei[pos] = whole ? sv[j] : edge_index[e];
                    sei[q + pos] = whole ? dv[j] : edge_index[Ecap + e];
                }
                if (sampled_p) sampled_p[pos] = p ? p[e] : 1.f;
            }
            bg += isgt;
            be += iseq;
        }
    }
    if (mask_aligned && e0 + kItems <= E) {
        *reinterpret_cast<uint64_t*>(mask + e0) = mbits;
    } else {
#pragma unroll
        for (int j = 0; j < kItems; ++j)
            if (e0 + j < E) mask[e0 + j] = static_cast<uint8_t>((mbits >> (8 * j)) & 1u);
    }
}
__global__ void __launch_bounds__(kThreads) compact(const uint32_t* __restrict__ keys, int64_t E, int64_t q, int64_t ties_override,
                                                   int64_t eid_offset, const SelectState* __restrict__ st, const uint2* __restrict__ cnt,
                                                   const float* __restrict__ p, const int64_t* __restrict__ edge_index,
                                                   uint8_t* __restrict__ mask, int mask_aligned,
                                                   int64_t* __restrict__ sampled_eid, int64_t* __restrict__ sei,
                                                   float* __restrict__ sampled_p) {
    compact_body(cnt[blockIdx.x].x, cnt[blockIdx.x].y, keys, E, E, q, ties_override, eid_offset, st, p, edge_index, mask, mask_aligned,
                 sampled_eid, sei, sampled_p);
}

// ---------------------------------------------------------------- fused small-E path (partition scale)
// At partition scale (E <= ~2 M) a draw is launch-latency bound, so sgs_sample_topq runs the same algorithm in
// 6 (learned) / 7 (prior) launches instead of 13 / 15.  The one-workgroup steps between the passes (final reductions,
// digit selection, block scan) are RECOMPUTED BY EVERY WORKGROUP of the next pass from the complete per-chunk partials /
// histograms / counts the previous launch left in memory -- a few KB of L2 reads per workgroup, the same arithmetic in
// the same order (bit-identical Z, max, threshold), no inter-workgroup synchronisation.  (A "last workgroup done"
// variant with tickets was measured slower on MI355X: the device-scope fences write back / invalidate the per-XCD L2s.)
constexpr int kSmallBlocks = 1024;
constexpr int kHistGrid = 2048;        // workgroups of the histogram passes on the large-E path (grid-stride over chunks)     // small path when E <= 1024 chunks (2 M candidate edges)

struct SelPart { uint32_t prefix, k_rem; };

// sgs_dyn_edges_set: the live candidate count comes from a device word; the launch was sized for the capacity.  Returns false
// for a workgroup that has no chunk of the live range (it leaves at once: it has no part in the recomputed reductions either).
__device__ __forceinline__ bool dyn_range(const int64_t* __restrict__ dynE, int64_t& E, int64_t& nblk) {
    if (dynE) {
        const int64_t live = *dynE;               // never more than the capacity the launch was sized for
        if (live < E) E = live;
        nblk = (E + kChunk - 1) / kChunk;
        return static_cast<int64_t>(blockIdx.x) < nblk;
    }
    return true;
}

template <int MODE>   // 1: max, otherwise sum (fixed tree, as reduce_final); result in every thread
__device__ __forceinline__ float final_reduce_all(const float* __restrict__ part, int64_t n, float* red) {
    float acc = (MODE == 1) ? -INFINITY : 0.f;
    // (the first kThreads threads only: a wider workgroup -- small_keys_hist0 -- reduces in exactly the order of a kThreads-wide one; the
    //  other waves contribute the identity to the fixed tree)
    if (threadIdx.x < kThreads)
        for (int64_t i = threadIdx.x; i < n; i += kThreads) {
            const float v = part[i];
            if (MODE == 1) acc = fmaxf(acc, v);
            else acc += v;
        }
    if (MODE == 1) return block_max(acc, red);
    const float r = block_sum(acc, red);
    __shared__ float bc;
    if (threadIdx.x == 0) bc = r;
    __syncthreads();
    const float out = bc;
    __syncthreads();
    return out;
}

// select_digit's search without its side effects: (bin, #keys in higher bins) of the k-th largest key; every thread
// gets the result.  `hist` is complete (previous launch).
__device__ __forceinline__ uint2 select_digit_local(const uint32_t* __restrict__ hist, uint32_t k) {
    __shared__ uint32_t tsum[kThreads];
    __shared__ uint32_t found_bin, found_above;
    constexpr int per = kBins / kThreads;
    uint32_t loc[per];
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < per; ++j) {
        loc[j] = hist[kBins - 1 - (threadIdx.x * per + j)];
        s += loc[j];
    }
    tsum[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < kThreads; off <<= 1) {
        uint32_t v = (threadIdx.x >= off) ? tsum[threadIdx.x - off] : 0u;
        __syncthreads();
        tsum[threadIdx.x] += v;
        __syncthreads();
    }
    const uint32_t before = tsum[threadIdx.x] - s;
    if (before < k && before + s >= k) {
        uint32_t run = before;
#pragma unroll
        for (int j = 0; j < per; ++j) {
            if (run < k && run + loc[j] >= k) {
                found_bin = kBins - 1 - (threadIdx.x * per + j);
                found_above = run;
            }
            run += loc[j];
        }
    }
    __syncthreads();
    const uint2 r = make_uint2(found_bin, found_above);
    __syncthreads();
    return r;
}

// first kernel of a draw: per-chunk partial (sum for LEARNED, max for PRIOR) + clears the three digit histograms
template <int MODE>
__global__ void __launch_bounds__(kThreads) small_reduce_first(const float* __restrict__ p, int64_t E, float* __restrict__ part,
                                                              uint32_t* __restrict__ hist3, const int64_t* __restrict__ dynE) {
    __shared__ float red[kThreads / 64];
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < 3 * kBins; i += gridDim.x * kThreads) hist3[i] = 0;
    int64_t nblk_ = 0;
    if (!dyn_range(dynE, E, nblk_)) return;
    const int64_t base = static_cast<int64_t>(blockIdx.x) * kChunk;
    float acc = (MODE == 1) ? -INFINITY : 0.f;
    // (loads first, unconditional on clamped indices, then the arithmetic: a load under `if (e < E)` with its use inside the branch is
    //  waited for inside the branch -- the unrolled items went through memory one after the other)
    float pv[kItems];
#pragma unroll
    for (int i = 0; i < kItems; ++i) {
        const int64_t e = base + static_cast<int64_t>(i) * kThreads + threadIdx.x;
        pv[i] = p ? p[e < E ? e : E - 1] : 1.f;
    }
#pragma unroll
    for (int i = 0; i < kItems; ++i) {
        const int64_t e = base + static_cast<int64_t>(i) * kThreads + threadIdx.x;
        if (e < E) {
            if (MODE == 1) acc = fmaxf(acc, pv[i]);
            else acc += pv[i];
        }
    }
    if (MODE == 1) {
        const float r = block_max(acc, red);
        if (threadIdx.x == 0) part[blockIdx.x] = r;
    } else {
        const float r = block_sum(acc, red);
        if (threadIdx.x == 0) part[blockIdx.x] = r;
    }
}

// PRIOR only: per-chunk partial of sum(exp(p - max)); the max is re-reduced by every workgroup from part_max
__global__ void __launch_bounds__(kThreads) small_reduce_sumexp(const float* __restrict__ p, int64_t E, const float* __restrict__ part_max,
                                                               int64_t nblk, float* __restrict__ part_sum, const int64_t* __restrict__ dynE) {
    __shared__ float red[kThreads / 64];
    if (!dyn_range(dynE, E, nblk)) return;
    const float mx = final_reduce_all<1>(part_max, nblk, red);
    const int64_t base = static_cast<int64_t>(blockIdx.x) * kChunk;
    float acc = 0.f;
    float pv[kItems];
#pragma unroll
    for (int i = 0; i < kItems; ++i) {
        const int64_t e = base + static_cast<int64_t>(i) * kThreads + threadIdx.x;
        pv[i] = p[e < E ? e : E - 1];
    }
#pragma unroll
    for (int i = 0; i < kItems; ++i) {
        const int64_t e = base + static_cast<int64_t>(i) * kThreads + threadIdx.x;
        if (e < E) acc += expf(pv[i] - mx);
    }
    const float r = block_sum(acc, red);
    if (threadIdx.x == 0) part_sum[blockIdx.x] = r;
}

// The key pass: 2 048 edges per workgroup as everywhere (one histogram flush and one redundant reduction per 2 048 keys), but 1 024 threads
// with TWO edges each.  With 256 threads x 8 edges (rounds 1-2) a partition's 172-242 workgroups put one wave on a SIMD, each running eight
// Philox draws back to back behind the reductions' round trips: 19 us under the counters, 65 % of it waiting (r03 PMC).  Sixteen waves per
// workgroup overlap those latencies.  (512-edge workgroups were tried in round 2: slower, four times the flushes and reductions.)
constexpr int kKeyThreads = 1024;
constexpr int kKeyItems = kChunk / kKeyThreads;
template <int MODE>
__global__ void __launch_bounds__(kKeyThreads) small_keys_hist0(const float* __restrict__ p, const float* __restrict__ prior,
                                                               const float* __restrict__ noise, uint64_t seed, uint64_t stream_id,
                                                               const uint64_t* __restrict__ epoch, int64_t E, float one_minus_c, float c,
                                                               const float* __restrict__ part_sum, const float* __restrict__ part_max,
                                                               int64_t nblk, float* __restrict__ scal, uint32_t* __restrict__ keys,
                                                               float* __restrict__ keys_out, uint32_t* __restrict__ hist0,
                                                               const int64_t* __restrict__ dynE) {
    __shared__ uint32_t lh[kBins];
    __shared__ float red[kKeyThreads / 64];
    if (!dyn_range(dynE, E, nblk)) return;
    for (int i = threadIdx.x; i < kBins; i += kKeyThreads) lh[i] = 0;
    seed = fold_epoch(seed, epoch);
    const bool has_prior = prior != nullptr;
    const int64_t base = static_cast<int64_t>(blockIdx.x) * kChunk;
    // every load of the thread in flight before the reductions' barriers and before the first use
    float pv[kKeyItems], qv[kKeyItems], nv[kKeyItems];
#pragma unroll
    for (int i = 0; i < kKeyItems; ++i) {
        const int64_t e = base + static_cast<int64_t>(i) * kKeyThreads + threadIdx.x;
        const int64_t ec = e < E ? e : E - 1;
        pv[i] = p ? p[ec] : 1.f;
        qv[i] = has_prior ? prior[ec] : 0.f;
        nv[i] = noise ? noise[ec] : 0.f;
    }
    const float mx = (MODE == SGS_SAMPLE_PRIOR) ? final_reduce_all<1>(part_max, nblk, red) : 0.f;
    const float Z = final_reduce_all<0>(part_sum, nblk, red);          // (syncs: lh is cleared for everyone)
    if (blockIdx.x == 0 && threadIdx.x == 0) { scal[0] = Z; scal[1] = mx; }
    const float Zeps = (MODE == SGS_SAMPLE_LEARNED) ? __fadd_rn(Z, 1e-12f) : Z;
    // all keys in registers BEFORE the first store: vmcnt counts stores too on this target, and behind per-item branches the compiler's
    // wait for an item's (long finished) loads was vmcnt(0) -- i.e. for the previous item's key store
    float kf[kKeyItems];
#pragma unroll
    for (int i = 0; i < kKeyItems; ++i) {
        const int64_t e = base + static_cast<int64_t>(i) * kKeyThreads + threadIdx.x;
        kf[i] = 0.f;
        if (e < E) {
            const float s = sample_prob<MODE>(pv[i], Zeps, mx, qv[i], has_prior, one_minus_c, c);
            const float nz = noise ? nv[i] : exp_noise_at(seed, stream_id, static_cast<uint64_t>(e));
            kf[i] = __fdiv_rn(s, nz);
        }
    }
#pragma unroll
    for (int i = 0; i < kKeyItems; ++i) {
        const int64_t e = base + static_cast<int64_t>(i) * kKeyThreads + threadIdx.x;
        if (e < E) {
            const uint32_t bits = __float_as_uint(kf[i]);
            keys[e] = bits;
            if (keys_out) keys_out[e] = kf[i];
            atomicAdd(&lh[bits >> kShift0], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kBins; i += kKeyThreads) {
        const uint32_t v = lh[i];
        if (v) atomicAdd(&hist0[i], v);
    }
}

// pass d (1 or 2): every workgroup first finishes digit d-1 from the complete hist_prev (+ the state workgroup 0 of the
// previous pass published), then histograms digit d of the keys matching the prefix; workgroup 0 publishes the state.
__global__ void __launch_bounds__(kThreads) small_hist_next(const uint32_t* __restrict__ keys, int64_t E, int shift, uint32_t digit_mask,
                                                           int prev_shift, int first, uint32_t q, const uint32_t* __restrict__ hist_prev,
                                                           const SelPart* __restrict__ sel_in, SelPart* __restrict__ sel_out,
                                                           uint32_t* __restrict__ hist, const int64_t* __restrict__ dynE) {
    __shared__ uint32_t lh[kBins];
    int64_t nblk_ = 0;
    if (!dyn_range(dynE, E, nblk_)) return;
    for (int i = threadIdx.x; i < kBins; i += kThreads) lh[i] = 0;
    const uint32_t k = first ? q : sel_in->k_rem;
    const uint32_t pre = first ? 0u : sel_in->prefix;
    const uint2 f = select_digit_local(hist_prev, k);                  // (syncs: lh is cleared for everyone)
    const uint32_t prefix = pre | (f.x << prev_shift);
    if (blockIdx.x == 0 && threadIdx.x == 0) { sel_out->prefix = prefix; sel_out->k_rem = k - f.y; }
    const uint32_t want = prefix >> prev_shift;
    const int64_t base = static_cast<int64_t>(blockIdx.x) * kChunk;
    uint32_t kv[kItems];
#pragma unroll
    for (int i = 0; i < kItems; ++i) {
        const int64_t e = base + static_cast<int64_t>(i) * kThreads + threadIdx.x;
        kv[i] = keys[e < E ? e : E - 1];
    }
#pragma unroll
    for (int i = 0; i < kItems; ++i) {
        const int64_t e = base + static_cast<int64_t>(i) * kThreads + threadIdx.x;
        if (e < E && (kv[i] >> prev_shift) == want) atomicAdd(&lh[(kv[i] >> shift) & digit_mask], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kBins; i += kThreads) {
        const uint32_t v = lh[i];
        if (v) atomicAdd(&hist[i], v);
    }
}

// every workgroup finishes the last digit (threshold T, ties to take), counts its chunk; workgroup 0 publishes the final
// SelectState and the stats
__global__ void __launch_bounds__(kThreads) small_count(const uint32_t* __restrict__ keys, int64_t E, uint32_t q,
                                                       const uint32_t* __restrict__ hist2, const SelPart* __restrict__ sel_in,
                                                       SelectState* __restrict__ st, uint2* __restrict__ cnt, const float* __restrict__ scal,
                                                       float* __restrict__ stats, const int64_t* __restrict__ dynE) {
    __shared__ int red[2 * (kThreads / 64)];
    int64_t nblk_ = 0;
    if (!dyn_range(dynE, E, nblk_)) return;
    const uint32_t k = sel_in->k_rem;
    const uint2 f = select_digit_local(hist2, k);
    const uint32_t T = sel_in->prefix | (f.x << kShift2);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->prefix = T;
        st->k_rem = k - f.y;
        st->n_gt = q - (k - f.y);
        st->pad = 0;
        if (stats) {
            stats[0] = scal[0];
            stats[1] = scal[1];
            stats[2] = __uint_as_float(T);
            stats[3] = static_cast<float>(k - f.y);
        }
    }
    const int64_t e0 = static_cast<int64_t>(blockIdx.x) * kChunk + static_cast<int64_t>(threadIdx.x) * kItems;
    uint32_t kk[kItems];
    load_keys8(keys, e0, E, kk);
    int gt = 0, eq = 0;
#pragma unroll
    for (int j = 0; j < kItems; ++j) {
        const bool in = e0 + j < E;
        gt += (in && kk[j] > T);
        eq += (in && kk[j] == T);
    }
    gt = wave_sum_int_all(gt);
    eq = wave_sum_int_all(eq);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) { red[2 * wid] = gt; red[2 * wid + 1] = eq; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int g = 0, q_ = 0;
        for (int w = 0; w < kThreads / 64; ++w) { g += red[2 * w]; q_ += red[2 * w + 1]; }
        cnt[blockIdx.x] = make_uint2(static_cast<uint32_t>(g), static_cast<uint32_t>(q_));
    }
}

// compaction with the exclusive prefix over the preceding chunks' (gt, eq) counts recomputed by each workgroup
__global__ void __launch_bounds__(kThreads) small_compact(const uint32_t* __restrict__ keys, int64_t E, int64_t q, const SelectState* __restrict__ st,
                                                         const uint2* __restrict__ cnt, const float* __restrict__ p,
                                                         const int64_t* __restrict__ edge_index, uint8_t* __restrict__ mask, int mask_aligned,
                                                         int64_t* __restrict__ sampled_eid, int64_t* __restrict__ sei,
                                                         float* __restrict__ sampled_p, const int64_t* __restrict__ dynE) {
    __shared__ int red[2 * (kThreads / 64)];
    __shared__ uint32_t pre[2];
    const int64_t Ecap = E;                       // row stride of edge_index (the capacity under sgs_dyn_edges_set)
    int64_t nblk_ = 0;
    if (!dyn_range(dynE, E, nblk_)) return;
    int g = 0, e = 0;
    for (int i = threadIdx.x; i < static_cast<int>(blockIdx.x); i += kThreads) { g += static_cast<int>(cnt[i].x); e += static_cast<int>(cnt[i].y); }
    g = wave_sum_int_all(g);
    e = wave_sum_int_all(e);
    if ((threadIdx.x & 63) == 0) { red[2 * (threadIdx.x >> 6)] = g; red[2 * (threadIdx.x >> 6) + 1] = e; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int G = 0, Q = 0;
        for (int w = 0; w < kThreads / 64; ++w) { G += red[2 * w]; Q += red[2 * w + 1]; }
        pre[0] = static_cast<uint32_t>(G);
        pre[1] = static_cast<uint32_t>(Q);
    }
    __syncthreads();
    compact_body(pre[0], pre[1], keys, E, Ecap, q, int64_t(-1), int64_t(0), st, p, edge_index, mask, mask_aligned, sampled_eid, sei, sampled_p);
}

// counts[0] = #keys > threshold, counts[1] = #keys == threshold in this shard (before the scan).
__global__ void __launch_bounds__(kThreads) total_counts(const uint2* __restrict__ cnt, int64_t nblk, uint32_t* __restrict__ counts) {
    __shared__ int red[2 * (kThreads / 64)];
    int g = 0, e = 0;
    for (int64_t i = threadIdx.x; i < nblk; i += kThreads) { g += static_cast<int>(cnt[i].x); e += static_cast<int>(cnt[i].y); }
    g = wave_sum_int_all(g);
    e = wave_sum_int_all(e);
    if ((threadIdx.x & 63) == 0) { red[2 * (threadIdx.x >> 6)] = g; red[2 * (threadIdx.x >> 6) + 1] = e; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int G = 0, Q = 0;
        for (int w = 0; w < kThreads / 64; ++w) { G += red[2 * w]; Q += red[2 * w + 1]; }
        counts[0] = static_cast<uint32_t>(G);
        counts[1] = static_cast<uint32_t>(Q);
    }
}

__global__ void write_stats(const float* __restrict__ scal, const SelectState* __restrict__ st, float* __restrict__ stats) {
    stats[0] = scal[0];
    stats[1] = scal[1];
    stats[2] = __uint_as_float(st->prefix);
    stats[3] = static_cast<float>(st->k_rem);
}

__global__ void select_all(int64_t E, const float* __restrict__ p, const int64_t* __restrict__ edge_index,
                           uint8_t* __restrict__ mask, int64_t* __restrict__ sampled_eid, int64_t* __restrict__ sei,
                           float* __restrict__ sampled_p, uint8_t value) {
    const int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (e >= E) return;
    mask[e] = value;
    if (value) {
        if (sampled_eid) sampled_eid[e] = e;
        if (sei) { sei[e] = edge_index[e]; sei[E + e] = edge_index[E + e]; }
        if (sampled_p) sampled_p[e] = p ? p[e] : 1.f;
    }
}

// out[0, j] = edge_index[0, idx[j]], out[1, j] = edge_index[1, idx[j]]  (sampling.py:161-163: edge_index[:, sampled_indices])
__global__ void gather_columns_kernel(const int64_t* __restrict__ edge_index, int64_t E, const int64_t* __restrict__ idx, int64_t q,
                                      int64_t* __restrict__ out) {
    const int64_t j = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (j >= q) return;
    const int64_t e = idx[j];
    const bool ok = e >= 0 && e < E;                  // out-of-range indices give (-1, -1) columns instead of a fault
    out[j] = ok ? edge_index[e] : int64_t(-1);
    out[q + j] = ok ? edge_index[E + e] : int64_t(-1);
}

__global__ void exp_noise_kernel(uint64_t seed, uint64_t stream_id, const uint64_t* __restrict__ epoch, int64_t E, float* __restrict__ noise) {
    seed = fold_epoch(seed, epoch);
    const int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (e < E) noise[e] = exp_noise_at(seed, stream_id, static_cast<uint64_t>(e));
}

__global__ void dropout_keep_kernel(uint64_t seed, uint32_t site, const uint64_t* __restrict__ epoch, int64_t rows, int64_t cols,
                                    uint32_t thresh, uint8_t* __restrict__ keep) {
    seed = fold_epoch(seed, epoch);
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < rows * cols) {
        const int64_t r = i / cols;
        const uint32_t c = static_cast<uint32_t>(i - r * cols);
        keep[i] = dropout_keep_at(seed, site, static_cast<uint64_t>(r), c, thresh) ? 1 : 0;
    }
}

// ---------------------------------------------------------------- straight-through weights
// sampling.py:137-138,155:  w = clamp(p * ((one_hot - s).detach() + s), 0, 1)[mask]
template <bool HAS_PRIOR>
__global__ void st_fwd_kernel(const float* __restrict__ p, const float* __restrict__ prior, float one_minus_c, float c,
                              const float* __restrict__ stats, const int64_t* __restrict__ eid, int64_t q,
                              float* __restrict__ w) {
    const int64_t j = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (j >= q) return;
    const int64_t e = eid[j];
    const float Zeps = __fadd_rn(stats[0], 1e-12f);
    const float pv = p[e];
    float s = __fdiv_rn(pv, Zeps);
    if (HAS_PRIOR) s = __fadd_rn(__fmul_rn(one_minus_c, s), __fmul_rn(c, prior[e]));
    const float st = __fadd_rn(__fsub_rn(1.0f, s), s);
    const float v = __fmul_rn(pv, st);
    w[j] = fminf(fmaxf(v, 0.f), 1.f);
}

// backward: h_e = g_e [0 <= p_e st_e <= 1];  S = sum_sel h_e p_e^2
//   dp_k = -(a/Z'^2) S  (all k)  +  [k sel] h_k (st_k + a p_k / Z'),  a = (1-c) or 1 (istest)
template <bool HAS_PRIOR>
__global__ void __launch_bounds__(kThreads) st_bwd_partial(const float* __restrict__ p, const float* __restrict__ prior,
                                                          float one_minus_c, float c, const float* __restrict__ stats,
                                                          const int64_t* __restrict__ eid, const float* __restrict__ gw,
                                                          int64_t q, float* __restrict__ part) {
    __shared__ float red[kThreads / 64];
    float acc = 0.f;
    const int64_t base = static_cast<int64_t>(blockIdx.x) * kChunk;
    const float Zeps = stats[0] + 1e-12f;
#pragma unroll
    for (int i = 0; i < kItems; ++i) {
        const int64_t j = base + static_cast<int64_t>(i) * kThreads + threadIdx.x;
        if (j < q) {
            const int64_t e = eid[j];
            const float pv = p[e];
            float s = pv / Zeps;
            if (HAS_PRIOR) s = one_minus_c * s + c * prior[e];
            const float v = pv * ((1.0f - s) + s);
            const float h = (v >= 0.f && v <= 1.f) ? gw[j] : 0.f;
            acc += h * pv * pv;
        }
    }
    const float r = block_sum(acc, red);
    if (threadIdx.x == 0) part[blockIdx.x] = r;
}

__global__ void __launch_bounds__(kThreads) st_bwd_final(const float* __restrict__ part, int64_t n, float* __restrict__ S) {
    __shared__ float red[kThreads / 64];
    float acc = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += kThreads) acc += part[i];
    const float r = block_sum(acc, red);
    if (threadIdx.x == 0) S[0] = r;
}

__global__ void st_bwd_dense(const float* __restrict__ S, const float* __restrict__ stats, float a, int64_t E,
                             float* __restrict__ dp, const int64_t* __restrict__ dynE) {
    const int64_t k = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (k >= E) return;
    const float Zeps = stats[0] + 1e-12f;
    // under sgs_dyn_edges_set the entries between the live count and the capacity are not edges: exactly zero gradient
    dp[k] = (dynE && k >= *dynE) ? 0.f : -(a / (Zeps * Zeps)) * S[0];

}

template <bool HAS_PRIOR>
__global__ void st_bwd_sparse(const float* __restrict__ p, const float* __restrict__ prior, float one_minus_c, float c,
                              float a, const float* __restrict__ stats, const int64_t* __restrict__ eid,
                              const float* __restrict__ gw, int64_t q, float* __restrict__ dp) {
    const int64_t j = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (j >= q) return;
    const int64_t e = eid[j];
    const float Zeps = stats[0] + 1e-12f;
    const float pv = p[e];
    float s = pv / Zeps;
    if (HAS_PRIOR) s = one_minus_c * s + c * prior[e];
    const float st = (1.0f - s) + s;
    const float v = pv * st;
    const float h = (v >= 0.f && v <= 1.f) ? gw[j] : 0.f;
    dp[e] += h * (st + a * pv / Zeps);   // eids are unique: no race
}

}  // namespace
}  // namespace sgs

using namespace sgs;

extern "C" {

int sgs_exp_noise(uint64_t seed, uint64_t stream_id, int64_t E, float* noise, sgs_stream_t stream) {
    SGS_REQUIRE(E >= 0 && (E == 0 || noise), SGS_EINVAL, "sgs_exp_noise: bad arguments");
    if (E == 0) return SGS_OK;
    hipLaunchKernelGGL(exp_noise_kernel, dim3(cdiv(E, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), seed,
                       stream_id, epoch_ptr(), E, noise);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

int sgs_dropout_keep(uint64_t seed, uint32_t site, int64_t rows, int64_t cols, float p, uint8_t* keep,
                     sgs_stream_t stream) {
    SGS_REQUIRE(rows >= 0 && cols >= 0 && cols < (int64_t(1) << 32) && p >= 0.f && p < 1.f, SGS_EINVAL,
                "sgs_dropout_keep: bad arguments");
    if (rows * cols == 0) return SGS_OK;
    SGS_REQUIRE(keep, SGS_EINVAL, "sgs_dropout_keep: null output");
    hipLaunchKernelGGL(dropout_keep_kernel, dim3(cdiv(rows * cols, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), seed, site, epoch_ptr(), rows, cols, dropout_thresh(p), keep);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

size_t sgs_sample_topq_workspace_bytes(int64_t E) {
    if (E < 0) E = 0;
    const int64_t nblk = cdiv(E, kChunk) + 1;
    return carve_bytes(E, 4)            // keys
           + carve_bytes(nblk, 4)       // partial sums
           + carve_bytes(4, 4)          // scalars Z, max
           + carve_bytes(kBins, 4)      // histogram
           + carve_bytes(1, sizeof(SelectState)) + carve_bytes(nblk, sizeof(uint2))
           + carve_bytes(nblk, 4) + carve_bytes(3 * kBins, 4) + 256 + 256;   // fused small-E path: second partials, per-digit histograms
}

int sgs_sample_topq(int mode, const float* p, const float* prior, double degree_bias_coef, const float* noise,
                    uint64_t seed, uint64_t stream_id, int64_t E, int64_t q, const int64_t* edge_index,
                    uint8_t* mask, int64_t* sampled_eid, int64_t* sampled_edge_index, float* sampled_p, float* stats,
                    float* keys_out, void* ws, size_t ws_bytes, sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(mode == SGS_SAMPLE_LEARNED || mode == SGS_SAMPLE_PRIOR, SGS_EINVAL, "sgs_sample_topq: bad mode %d", mode);
    SGS_REQUIRE(E >= 0 && q >= 0, SGS_EINVAL, "sgs_sample_topq: negative size (E=%lld q=%lld)", (long long)E, (long long)q);
    SGS_REQUIRE(q <= E, SGS_EINVAL,
                "sgs_sample_topq: cannot sample q=%lld > E=%lld edges without replacement", (long long)q, (long long)E);
    SGS_REQUIRE(E < (int64_t(1) << 32), SGS_EINVAL, "sgs_sample_topq: E=%lld exceeds 2^32-1", (long long)E);
    if (E == 0) return SGS_OK;
    SGS_REQUIRE(mask, SGS_EINVAL, "sgs_sample_topq: null mask");
    SGS_REQUIRE(p || (mode == SGS_SAMPLE_LEARNED && !prior), SGS_EINVAL, "sgs_sample_topq: p == NULL (uniform weights) needs mode LEARNED and no prior");
    SGS_REQUIRE(!sampled_edge_index || edge_index, SGS_EINVAL, "sgs_sample_topq: edge_index required for sampled_edge_index");
    SGS_REQUIRE(ws && ws_bytes >= sgs_sample_topq_workspace_bytes(E), SGS_EWORKSPACE,
                "sgs_sample_topq: workspace too small (%zu < %zu)", ws_bytes, sgs_sample_topq_workspace_bytes(E));
    SGS_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 255) == 0, SGS_EINVAL, "sgs_sample_topq: workspace must be 256-B aligned");

    const int64_t nblk = cdiv(E, kChunk);
    Carver cv(ws);
    uint32_t* keys = cv.take<uint32_t>(E);
    float* part = cv.take<float>(nblk + 1);
    float* scal = cv.take<float>(4);
    uint32_t* hist = cv.take<uint32_t>(kBins);
    SelectState* st = cv.take<SelectState>(1);
    uint2* cnt = cv.take<uint2>(nblk + 1);

    float* part2 = cv.take<float>(nblk + 1);
    uint32_t* hist3 = cv.take<uint32_t>(3 * kBins);       // fused small-E path: one histogram per digit
    SelPart* sel = cv.take<SelPart>(2);
    const dim3 grid(static_cast<unsigned>(nblk)), blk(kThreads);
    const dim3 hgrid(static_cast<unsigned>(nblk < kHistGrid ? nblk : kHistGrid));
    // python: (1 - c) and c are doubles, cast to fp32 when they meet the fp32 tensor
    const float one_minus_c = static_cast<float>(1.0 - degree_bias_coef);
    const float c = static_cast<float>(degree_bias_coef);
    const int mask_aligned = (reinterpret_cast<uintptr_t>(mask) & 7) == 0;

    const int64_t* dynE = dyn_edges_ptr();
    SGS_REQUIRE(!dynE || (nblk <= kSmallBlocks && q > 0 && q < E), SGS_EINVAL,
                "sgs_sample_topq: a dynamic edge count (sgs_dyn_edges_set) needs 0 < q < capacity <= %lld", (long long)kSmallBlocks * kChunk);
    if (nblk <= kSmallBlocks && q > 0 && q < E) {
        // fused small-E path: 6 / 7 launches (see "fused small-E path" above); same arithmetic, same results
        const uint32_t q32 = static_cast<uint32_t>(q);
        uint32_t *h0 = hist3, *h1 = hist3 + kBins, *h2 = hist3 + 2 * kBins;
        const dim3 kgrid(static_cast<unsigned>(nblk)), kblk(kKeyThreads);
        if (mode == SGS_SAMPLE_LEARNED) {
            hipLaunchKernelGGL(small_reduce_first<0>, grid, blk, 0, stream, p, E, part, hist3, dynE);
            hipLaunchKernelGGL(small_keys_hist0<SGS_SAMPLE_LEARNED>, kgrid, kblk, 0, stream, p, prior, noise, seed, stream_id, epoch_ptr(), E,
                               one_minus_c, c, part, static_cast<const float*>(nullptr), nblk, scal, keys, keys_out, h0, dynE);
        } else {
            hipLaunchKernelGGL(small_reduce_first<1>, grid, blk, 0, stream, p, E, part, hist3, dynE);
            hipLaunchKernelGGL(small_reduce_sumexp, grid, blk, 0, stream, p, E, part, nblk, part2, dynE);
            hipLaunchKernelGGL(small_keys_hist0<SGS_SAMPLE_PRIOR>, kgrid, kblk, 0, stream, p, static_cast<const float*>(nullptr), noise, seed,
                               stream_id, epoch_ptr(), E, one_minus_c, c, part2, part, nblk, scal, keys, keys_out, h0, dynE);
        }
        hipLaunchKernelGGL(small_hist_next, grid, blk, 0, stream, keys, E, kShift1, kMask1, kShift0, 1, q32, h0,
                           static_cast<const SelPart*>(nullptr), sel, h1, dynE);
        hipLaunchKernelGGL(small_hist_next, grid, blk, 0, stream, keys, E, kShift2, kMask2, kShift1, 0, q32, h1, sel, sel + 1, h2, dynE);
        hipLaunchKernelGGL(small_count, grid, blk, 0, stream, keys, E, q32, h2, sel + 1, st, cnt, scal, stats, dynE);
        hipLaunchKernelGGL(small_compact, grid, blk, 0, stream, keys, E, q, st, cnt, p, edge_index, mask, mask_aligned, sampled_eid,
                           sampled_edge_index, sampled_p, dynE);
        SGS_LAUNCH_OK();
        return SGS_OK;
    }

    // scal, hist and st are adjacent carvings: one zeroing launch (a kernel, not a memset node: see zero_async)
    if (int rc = zero_async(scal, static_cast<size_t>(reinterpret_cast<char*>(st + 1) - reinterpret_cast<char*>(scal)), stream)) return rc;

    if (mode == SGS_SAMPLE_LEARNED) {
        hipLaunchKernelGGL(reduce_partial<0>, grid, blk, 0, stream, p, E, scal, part);
        hipLaunchKernelGGL(reduce_final<0>, dim3(1), blk, 0, stream, part, nblk, scal);
    } else {
        hipLaunchKernelGGL(reduce_partial<1>, grid, blk, 0, stream, p, E, scal, part);
        hipLaunchKernelGGL(reduce_final<1>, dim3(1), blk, 0, stream, part, nblk, scal);
        hipLaunchKernelGGL(reduce_partial<2>, grid, blk, 0, stream, p, E, scal, part);
        hipLaunchKernelGGL(reduce_final<2>, dim3(1), blk, 0, stream, part, nblk, scal);
    }
    SGS_LAUNCH_OK();

    if (q == 0 || q == E) {   // degenerate draws: nothing / everything
        hipLaunchKernelGGL(select_all, dim3(cdiv(E, 256)), dim3(256), 0, stream, E, p, edge_index, mask, sampled_eid,
                           sampled_edge_index, sampled_p, static_cast<uint8_t>(q == E ? 1 : 0));
        if (stats) hipLaunchKernelGGL(write_stats, dim3(1), dim3(1), 0, stream, scal, st, stats);
        SGS_LAUNCH_OK();
        return SGS_OK;
    }

    if (mode == SGS_SAMPLE_LEARNED)
        hipLaunchKernelGGL(keys_hist0<SGS_SAMPLE_LEARNED>, hgrid, blk, 0, stream, p, prior, noise, seed, stream_id, epoch_ptr(), int64_t(0), E,
                           one_minus_c, c, scal, keys, keys_out, hist);
    else
        hipLaunchKernelGGL(keys_hist0<SGS_SAMPLE_PRIOR>, hgrid, blk, 0, stream, p, nullptr, noise, seed, stream_id, epoch_ptr(), int64_t(0), E,
                           one_minus_c, c, scal, keys, keys_out, hist);
    hipLaunchKernelGGL(select_digit, dim3(1), blk, 0, stream, hist, kShift0, 1, static_cast<uint32_t>(q), st);
    hipLaunchKernelGGL(hist_next, hgrid, blk, 0, stream, keys, E, kShift1, kMask1, kShift0, st, hist);
    hipLaunchKernelGGL(select_digit, dim3(1), blk, 0, stream, hist, kShift1, 0, static_cast<uint32_t>(q), st);
    hipLaunchKernelGGL(hist_next, hgrid, blk, 0, stream, keys, E, kShift2, kMask2, kShift1, st, hist);
    hipLaunchKernelGGL(select_digit, dim3(1), blk, 0, stream, hist, kShift2, 0, static_cast<uint32_t>(q), st);
    hipLaunchKernelGGL(count_blocks, grid, blk, 0, stream, keys, E, st, cnt);
    hipLaunchKernelGGL(scan_blocks, dim3(1), blk, 0, stream, cnt, nblk);
    hipLaunchKernelGGL(compact, grid, blk, 0, stream, keys, E, q, int64_t(-1), int64_t(0), st, cnt, p, edge_index, mask, mask_aligned,
                       sampled_eid, sampled_edge_index, sampled_p);
    if (stats) hipLaunchKernelGGL(write_stats, dim3(1), dim3(1), 0, stream, scal, st, stats);
    SGS_LAUNCH_OK();
    return SGS_OK;
}


/* ---------------------------------------------------------------- phase API (edge-sharded draws)
 * The same kernels as sgs_sample_topq, exposed per phase so that R ranks holding contiguous,
 * 2048-aligned shards of the edge list can run ONE exact global draw: per-chunk partial sums are
 * all-gathered and reduced in the single-GPU order (bit-identical Z), digit histograms are
 * all-reduced (integers), every rank runs the same digit selection, ties at the threshold go to the
 * lowest GLOBAL edge ids, and compaction is local.  Noise is keyed by the global edge id. */
size_t sgs_sampler_shard_workspace_bytes(int64_t E_local) {
    if (E_local < 0) E_local = 0;
    return carve_bytes(E_local, 4) + carve_bytes(cdiv(E_local, kChunk) + 1, sizeof(uint2)) + 256;
}
int64_t sgs_sampler_chunk(void) { return kChunk; }

int sgs_sampler_shard_partials(int stage, const float* p, int64_t E, const float* scal, float* part, sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(stage >= 0 && stage <= 2 && E >= 0, SGS_EINVAL, "sgs_sampler_shard_partials: bad arguments");
    if (E == 0) return SGS_OK;
    SGS_REQUIRE(p && part && (stage != 2 || scal), SGS_EINVAL, "sgs_sampler_shard_partials: null pointer");
    const dim3 grid(static_cast<unsigned>(cdiv(E, kChunk))), blk(kThreads);
    if (stage == 0) hipLaunchKernelGGL(reduce_partial<0>, grid, blk, 0, stream, p, E, scal, part);
    else if (stage == 1) hipLaunchKernelGGL(reduce_partial<1>, grid, blk, 0, stream, p, E, scal, part);
    else hipLaunchKernelGGL(reduce_partial<2>, grid, blk, 0, stream, p, E, scal, part);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

int sgs_sampler_shard_finalize(int stage, const float* part_all, int64_t nblk_all, float* scal, sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(stage >= 0 && stage <= 2 && nblk_all >= 0 && scal && (nblk_all == 0 || part_all), SGS_EINVAL,
                "sgs_sampler_shard_finalize: bad arguments");
    if (stage == 1) hipLaunchKernelGGL(reduce_final<1>, dim3(1), dim3(kThreads), 0, stream, part_all, nblk_all, scal);
    else hipLaunchKernelGGL(reduce_final<0>, dim3(1), dim3(kThreads), 0, stream, part_all, nblk_all, scal);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

int sgs_sampler_shard_keys(int mode, const float* p, const float* prior, double degree_bias_coef, const float* noise,
                           uint64_t seed, uint64_t stream_id, int64_t edge_offset, int64_t E, const float* scal, uint32_t* keys,
                           float* keys_out, uint32_t* hist, sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE((mode == SGS_SAMPLE_LEARNED || mode == SGS_SAMPLE_PRIOR) && E >= 0 && edge_offset >= 0, SGS_EINVAL,
                "sgs_sampler_shard_keys: bad arguments");
    if (E == 0) return SGS_OK;
    SGS_REQUIRE(p && scal && keys && hist, SGS_EINVAL, "sgs_sampler_shard_keys: null pointer");
    const float one_minus_c = static_cast<float>(1.0 - degree_bias_coef), c = static_cast<float>(degree_bias_coef);
    const int64_t nb_ = cdiv(E, kChunk);
    const dim3 hgrid(static_cast<unsigned>(nb_ < kHistGrid ? nb_ : kHistGrid)), blk(kThreads);
    if (mode == SGS_SAMPLE_LEARNED)
        hipLaunchKernelGGL(keys_hist0<SGS_SAMPLE_LEARNED>, hgrid, blk, 0, stream, p, prior, noise, seed, stream_id, epoch_ptr(), edge_offset, E,
                           one_minus_c, c, scal, keys, keys_out, hist);
    else
        hipLaunchKernelGGL(keys_hist0<SGS_SAMPLE_PRIOR>, hgrid, blk, 0, stream, p, nullptr, noise, seed, stream_id, epoch_ptr(), edge_offset, E,
                           one_minus_c, c, scal, keys, keys_out, hist);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

int sgs_sampler_shard_hist(const uint32_t* keys, int64_t E, int pass, const void* state, uint32_t* hist, sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE((pass == 1 || pass == 2) && E >= 0 && state && hist, SGS_EINVAL, "sgs_sampler_shard_hist: bad arguments");
    if (E == 0) return SGS_OK;
    const int64_t nb_ = cdiv(E, kChunk);
    const dim3 hgrid(static_cast<unsigned>(nb_ < kHistGrid ? nb_ : kHistGrid)), blk(kThreads);
    if (pass == 1) hipLaunchKernelGGL(hist_next, hgrid, blk, 0, stream, keys, E, kShift1, kMask1, kShift0, static_cast<const SelectState*>(state), hist);
    else hipLaunchKernelGGL(hist_next, hgrid, blk, 0, stream, keys, E, kShift2, kMask2, kShift1, static_cast<const SelectState*>(state), hist);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

int sgs_sampler_select(uint32_t* hist, int pass, int64_t q, void* state, sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(pass >= 0 && pass <= 2 && q > 0 && hist && state, SGS_EINVAL, "sgs_sampler_select: bad arguments");
    const int shift = pass == 0 ? kShift0 : pass == 1 ? kShift1 : kShift2;
    hipLaunchKernelGGL(select_digit, dim3(1), dim3(kThreads), 0, stream, hist, shift, pass == 0 ? 1 : 0, static_cast<uint32_t>(q),
                       static_cast<SelectState*>(state));
    SGS_LAUNCH_OK();
    return SGS_OK;
}

int sgs_sampler_shard_count(const uint32_t* keys, int64_t E, const void* state, uint32_t* counts, void* ws, size_t ws_bytes,
                            sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(E >= 0 && state && counts, SGS_EINVAL, "sgs_sampler_shard_count: bad arguments");
    SGS_REQUIRE(ws && ws_bytes >= sgs_sampler_shard_workspace_bytes(E), SGS_EWORKSPACE, "sgs_sampler_shard_count: workspace too small");
    Carver cv(ws);
    cv.take<uint32_t>(E);                                  // keys live in the caller's copy of this region
    uint2* cnt = cv.take<uint2>(cdiv(E, kChunk) + 1);
    const int64_t nblk = cdiv(E, kChunk);
    if (nblk > 0) hipLaunchKernelGGL(count_blocks, dim3(nblk), dim3(kThreads), 0, stream, keys, E, static_cast<const SelectState*>(state), cnt);
    hipLaunchKernelGGL(total_counts, dim3(1), dim3(kThreads), 0, stream, cnt, nblk, counts);
    if (nblk > 0) hipLaunchKernelGGL(scan_blocks, dim3(1), dim3(kThreads), 0, stream, cnt, nblk);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

int sgs_sampler_shard_compact(const uint32_t* keys, int64_t E, const void* state, int64_t ties_local, int64_t q_local,
                              int64_t edge_offset, const float* p, const int64_t* edge_index_local, uint8_t* mask,
                              int64_t* sampled_eid, int64_t* sampled_edge_index, float* sampled_p, void* ws, size_t ws_bytes,
                              sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(E >= 0 && ties_local >= 0 && q_local >= 0 && state, SGS_EINVAL, "sgs_sampler_shard_compact: bad arguments");
    if (E == 0) return SGS_OK;
    SGS_REQUIRE(keys && mask && p, SGS_EINVAL, "sgs_sampler_shard_compact: null pointer");
    SGS_REQUIRE(ws && ws_bytes >= sgs_sampler_shard_workspace_bytes(E), SGS_EWORKSPACE, "sgs_sampler_shard_compact: workspace too small");
    Carver cv(ws);
    cv.take<uint32_t>(E);
    uint2* cnt = cv.take<uint2>(cdiv(E, kChunk) + 1);      // scanned by sgs_sampler_shard_count
    const int mask_aligned = (reinterpret_cast<uintptr_t>(mask) & 7) == 0;
    hipLaunchKernelGGL(compact, dim3(cdiv(E, kChunk)), dim3(kThreads), 0, stream, keys, E, q_local, ties_local, edge_offset,
                       static_cast<const SelectState*>(state), cnt, p, edge_index_local, mask, mask_aligned, sampled_eid,
                       sampled_edge_index, sampled_p);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

int sgs_gather_columns(const int64_t* edge_index, int64_t E, const int64_t* idx, int64_t q, int64_t* out, sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(E >= 0 && q >= 0, SGS_EINVAL, "sgs_gather_columns: bad sizes");
    if (q == 0) return SGS_OK;
    SGS_REQUIRE(edge_index && idx && out, SGS_EINVAL, "sgs_gather_columns: null pointer");
    hipLaunchKernelGGL(gather_columns_kernel, dim3(cdiv(q, 256)), dim3(256), 0, stream, edge_index, E, idx, q, out);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

int sgs_st_weights_fwd(const float* p, const float* prior, double degree_bias_coef, const float* stats,
                       const int64_t* sampled_eid, int64_t E, int64_t q, float* w, sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(E >= 0 && q >= 0 && q <= E, SGS_EINVAL, "sgs_st_weights_fwd: bad sizes");
    if (q == 0) return SGS_OK;
    SGS_REQUIRE(p && stats && sampled_eid && w, SGS_EINVAL, "sgs_st_weights_fwd: null pointer");
    const float one_minus_c = static_cast<float>(1.0 - degree_bias_coef), c = static_cast<float>(degree_bias_coef);
    if (prior)
        hipLaunchKernelGGL(st_fwd_kernel<true>, dim3(cdiv(q, 256)), dim3(256), 0, stream, p, prior, one_minus_c, c, stats,
                           sampled_eid, q, w);
    else
        hipLaunchKernelGGL(st_fwd_kernel<false>, dim3(cdiv(q, 256)), dim3(256), 0, stream, p, prior, one_minus_c, c, stats,
                           sampled_eid, q, w);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

size_t sgs_st_weights_bwd_workspace_bytes(int64_t E, int64_t q) {
    (void)E;
    if (q < 0) q = 0;
    return carve_bytes(cdiv(q, kChunk) + 1, 4) + carve_bytes(4, 4) + 256;
}

int sgs_st_weights_bwd(const float* p, const float* prior, double degree_bias_coef, const float* stats,
                       const int64_t* sampled_eid, const float* grad_w, int64_t E, int64_t q, float* grad_p, void* ws,
                       size_t ws_bytes, sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(E >= 0 && q >= 0 && q <= E, SGS_EINVAL, "sgs_st_weights_bwd: bad sizes");
    if (E == 0) return SGS_OK;
    SGS_REQUIRE(p && stats && grad_p && (q == 0 || (sampled_eid && grad_w)), SGS_EINVAL, "sgs_st_weights_bwd: null pointer");
    SGS_REQUIRE(ws && ws_bytes >= sgs_st_weights_bwd_workspace_bytes(E, q), SGS_EWORKSPACE,
                "sgs_st_weights_bwd: workspace too small");
    Carver cv(ws);
    const int64_t nblk = cdiv(q, kChunk);
    float* part = cv.take<float>(nblk + 1);
    float* S = cv.take<float>(4);
    const float one_minus_c = static_cast<float>(1.0 - degree_bias_coef), c = static_cast<float>(degree_bias_coef);
    const float a = prior ? one_minus_c : 1.0f;
    if (int rc = zero_async(S, 16, stream)) return rc;
    if (q > 0) {
        if (prior)
            hipLaunchKernelGGL(st_bwd_partial<true>, dim3(nblk), dim3(kThreads), 0, stream, p, prior, one_minus_c, c, stats,
                               sampled_eid, grad_w, q, part);
        else
            hipLaunchKernelGGL(st_bwd_partial<false>, dim3(nblk), dim3(kThreads), 0, stream, p, prior, one_minus_c, c, stats,
                               sampled_eid, grad_w, q, part);
        hipLaunchKernelGGL(st_bwd_final, dim3(1), dim3(kThreads), 0, stream, part, nblk, S);
    }
    hipLaunchKernelGGL(st_bwd_dense, dim3(cdiv(E, 256)), dim3(256), 0, stream, S, stats, a, E, grad_p, dyn_edges_ptr());
    if (q > 0) {
        if (prior)
            hipLaunchKernelGGL(st_bwd_sparse<true>, dim3(cdiv(q, 256)), dim3(256), 0, stream, p, prior, one_minus_c, c, a,
                               stats, sampled_eid, grad_w, q, grad_p);
        else
            hipLaunchKernelGGL(st_bwd_sparse<false>, dim3(cdiv(q, 256)), dim3(256), 0, stream, p, prior, one_minus_c, c, a,
                               stats, sampled_eid, grad_w, q, grad_p);
    }
    SGS_LAUNCH_OK();
    return SGS_OK;
}

}  // extern "C"
