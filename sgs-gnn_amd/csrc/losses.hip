// K6: gate and losses of the hybrid / straight-through / two-pass step (gfx950).
//
// Reference: training_hybrid.py:92-133 and utils.py:163-169 (`calculate_f1`), 187-211
// (`consistency_loss`).  The reference copies logits to the host for sklearn's micro-F1, runs
// two sort-based torch.isin calls and one .item() sync for reg1; here everything stays on the
// device and every reduction uses a fixed tree (deterministic).
#include "sgs_common.h"

namespace sgs {
namespace {

constexpr int kT = 256;

// ---------------------------------------------------------------- gate: #correct argmax on train rows
// One wave per row; torch.argmax semantics (first maximum wins).
__global__ void __launch_bounds__(kT) masked_correct(const float* __restrict__ logits, int64_t N, int64_t C,
                                                    const int64_t* __restrict__ y, const uint8_t* __restrict__ mask,
                                                    int* __restrict__ correct) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x) >> 6;
    if (i >= N || !mask[i]) return;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int64_t c = lane; c < C; c += 64) {
        const float v = logits[i * C + c];
        if (v > best || (v == best && static_cast<int>(c) < bi) || bi == 0x7fffffff) { best = v; bi = static_cast<int>(c); }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > best || (ov == best && oi < bi))) { best = ov; bi = oi; }
    }
    if (lane == 0) {
        atomicAdd(&correct[1], 1);
        if (static_cast<int64_t>(bi) == y[i]) atomicAdd(&correct[0], 1);
    }
}

// The gate's two counts in one launch: waves [0, N) score logits_a into correct[0:2], waves [N, 2N) logits_b into
// correct[2:4].  `correct` must be zero on entry (the caller's allocation clears it: one fill for both).
__global__ void __launch_bounds__(1024) masked_correct_pair(const float* __restrict__ logits_a, const float* __restrict__ logits_b, int64_t N,
                                                           int64_t C, const int64_t* __restrict__ y, const uint8_t* __restrict__ mask,
                                                           int* __restrict__ correct) {
    // 16 waves = 16 rows of ONE of the two logit matrices per workgroup (blockIdx.y selects it): the counts are reduced in LDS
    // and each workgroup issues two atomics instead of every row issuing two
    __shared__ int red[2 * 16];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 16 + wid;
    const float* __restrict__ logits = blockIdx.y ? logits_b : logits_a;
    int ok = 0, row = 0;
    if (i < N && mask[i]) {
        float best = -INFINITY;
        int bi = 0x7fffffff;
        for (int64_t c = lane; c < C; c += 64) {
            const float v = logits[i * C + c];
            if (v > best || (v == best && static_cast<int>(c) < bi) || bi == 0x7fffffff) { best = v; bi = static_cast<int>(c); }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > best || (ov == best && oi < bi))) { best = ov; bi = oi; }
        }
        row = 1;
        ok = (static_cast<int64_t>(bi) == y[i]) ? 1 : 0;
    }
    if (lane == 0) { red[2 * wid] = ok; red[2 * wid + 1] = row; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int a_ = 0, b_ = 0;
        for (int w = 0; w < 16; ++w) { a_ += red[2 * w]; b_ += red[2 * w + 1]; }
        int* out = correct + (blockIdx.y ? 2 : 0);
        if (b_) atomicAdd(&out[1], b_);
        if (a_) atomicAdd(&out[0], a_);
    }
}

// The gate in two launches with no zero fill and no atomics: gate_counts_partial = masked_correct_pair's row scan, but every workgroup
// writes its own (#correct, #rows) pair; gate_counts_finish (one wave) adds them in order, writes out[0..3] = (#correct_a, #train,
// #correct_b, #train), out[4] = 0 and -- with `dst` -- hands the four counts to the polling host exactly as publish_words does.
__global__ void __launch_bounds__(1024) gate_counts_partial(const float* __restrict__ logits_a, const float* __restrict__ logits_b, int64_t N,
                                                           int64_t C, const int64_t* __restrict__ y, const uint8_t* __restrict__ mask,
                                                           int2* __restrict__ part) {
    __shared__ int red[2 * 16];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 16 + wid;
    const float* __restrict__ logits = blockIdx.y ? logits_b : logits_a;
    int ok = 0, row = 0;
    if (i < N && mask[i]) {
        float best = -INFINITY;
        int bi = 0x7fffffff;
        for (int64_t c = lane; c < C; c += 64) {
            const float v = logits[i * C + c];
            if (v > best || (v == best && static_cast<int>(c) < bi) || bi == 0x7fffffff) { best = v; bi = static_cast<int>(c); }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > best || (ov == best && oi < bi))) { best = ov; bi = oi; }
        }
        row = 1;
        ok = (static_cast<int64_t>(bi) == y[i]) ? 1 : 0;
    }
    if (lane == 0) { red[2 * wid] = ok; red[2 * wid + 1] = row; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int a_ = 0, b_ = 0;
        for (int w = 0; w < 16; ++w) { a_ += red[2 * w]; b_ += red[2 * w + 1]; }
        part[static_cast<int64_t>(blockIdx.y) * gridDim.x + blockIdx.x] = make_int2(a_, b_);
    }
}
__global__ void __launch_bounds__(64) gate_counts_finish(const int2* __restrict__ part, int nblk, int* __restrict__ out,
                                                        const uint64_t* __restrict__ seq, int32_t* __restrict__ dst) {
    int a0 = 0, r0 = 0, a1 = 0, r1 = 0;
    for (int b = threadIdx.x; b < nblk; b += 64) {
        const int2 pa = part[b], pb = part[nblk + b];
        a0 += pa.x; r0 += pa.y; a1 += pb.x; r1 += pb.y;
    }
    a0 = wave_sum_int_all(a0); r0 = wave_sum_int_all(r0); a1 = wave_sum_int_all(a1); r1 = wave_sum_int_all(r1);
    if (threadIdx.x == 0) {
        out[0] = a0; out[1] = r0; out[2] = a1; out[3] = r1; out[4] = 0;
        if (dst) {
            __hip_atomic_store(dst + 0, a0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(dst + 1, r0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(dst + 2, a1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(dst + 3, r1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __atomic_thread_fence(__ATOMIC_RELEASE);
            __hip_atomic_store(dst + 4, static_cast<int32_t>(seq ? seq[0] : 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// End of a replayed step in one launch: running loss += this step's loss, RNG epoch += 1 (the next replay draws fresh noise).
__global__ void loss_tick(float* __restrict__ sum, const float* __restrict__ loss, uint64_t* __restrict__ epoch) {
    if (threadIdx.x == 0) {
        if (sum && loss) sum[0] += loss[0];
        if (epoch) epoch[0] += 1;
    }
}

// Hands a few device words to the HOST without a copy engine round trip: `dst` is pinned, device-mapped host memory; the
// payload is written first, then the sequence word (the RNG epoch, which changes on every graph replay) with release
// semantics at system scope, so a host thread that polls the sequence word sees the payload.  Used for the gate.
__global__ void publish_words(const int32_t* __restrict__ src, int n, const uint64_t* __restrict__ seq, int32_t* dst) {
    if (threadIdx.x < n) __hip_atomic_store(dst + threadIdx.x, src[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __syncthreads();
    if (threadIdx.x == 0) {
        __atomic_thread_fence(__ATOMIC_RELEASE);
        __hip_atomic_store(dst + n, static_cast<int32_t>(seq ? seq[0] : 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---------------------------------------------------------------- masked cross entropy
// rowloss[i] = lse_i - logit_i[y_i] on train rows (0 elsewhere); row_lse kept for backward.
__global__ void __launch_bounds__(kT) ce_rows(const float* __restrict__ logits, int64_t N, int64_t C, const int64_t* __restrict__ y,
                                             const uint8_t* __restrict__ mask, float* __restrict__ row_lse,
                                             float* __restrict__ rowloss) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x) >> 6;
    if (i >= N) return;
    if (!mask[i]) {
        if (lane == 0) { rowloss[i] = 0.f; row_lse[i] = 0.f; }
        return;
    }
    float mx = -INFINITY;
    for (int64_t c = lane; c < C; c += 64) mx = fmaxf(mx, logits[i * C + c]);
    mx = wave_max_all(mx);
    float s = 0.f;
    for (int64_t c = lane; c < C; c += 64) s += expf(logits[i * C + c] - mx);
    s = wave_sum_all(s);
    const float lse = mx + logf(s);
    if (lane == 0) {
        row_lse[i] = lse;
        rowloss[i] = lse - logits[i * C + y[i]];
    }
}

// One block: loss = sum(rowloss) / #train ; n_rows[0] = #train.
__global__ void __launch_bounds__(1024) ce_final(const float* __restrict__ rowloss, const uint8_t* __restrict__ mask, int64_t N,
                                                float* __restrict__ loss, int* __restrict__ n_rows) {
    __shared__ float red[16];
    __shared__ int redi[16];
    float acc = 0.f;
    int cnt = 0;
    for (int64_t i = threadIdx.x; i < N; i += 1024) { acc += rowloss[i]; cnt += mask[i] ? 1 : 0; }
    const float r = block_sum(acc, red);
    cnt = wave_sum_int_all(cnt);
    if ((threadIdx.x & 63) == 0) redi[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        int n = 0;
        for (int w = 0; w < 16; ++w) n += redi[w];
        n_rows[0] = n;
        loss[0] = r / static_cast<float>(n);     // 0/0 = nan, as torch's mean over an empty selection
    }
}

__global__ void __launch_bounds__(kT) ce_bwd(const float* __restrict__ logits, int64_t N, int64_t C, const int64_t* __restrict__ y,
                                            const uint8_t* __restrict__ mask, const float* __restrict__ row_lse,
                                            const int* __restrict__ n_rows, const float* __restrict__ grad_loss,
                                            float* __restrict__ dlogits) {
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x;
    if (idx >= N * C) return;
    const int64_t i = idx / C, c = idx - i * C;
    float g = 0.f;
    if (mask[i]) {
        const float sm = expf(logits[idx] - row_lse[i]);
        g = (sm - (y[i] == c ? 1.f : 0.f)) * (grad_loss[0] / static_cast<float>(n_rows[0]));
    }
    dlogits[idx] = g;
}

// ---------------------------------------------------------------- edge regularisers
struct EdgeTerms {
    float dot, nx, ny;   // <x,y>, |x|^2, |y|^2
};
__device__ __forceinline__ EdgeTerms edge_dot(const float* __restrict__ x, const float* __restrict__ y, int64_t C) {
    EdgeTerms t{0.f, 0.f, 0.f};
    for (int64_t c = 0; c < C; ++c) {
        const float a = x[c], b = y[c];
        t.dot = fmaf(a, b, t.dot); t.nx = fmaf(a, a, t.nx); t.ny = fmaf(b, b, t.ny);
    }
    return t;
}
// F.cosine_similarity(x, y, eps=1e-8) = <x,y> / sqrt(max(|x|^2 |y|^2, eps^2))
__device__ __forceinline__ float cos_from(const EdgeTerms& t) { return t.dot / sqrtf(fmaxf(t.nx * t.ny, 1e-16f)); }

// edges per workgroup of the forward pass: kRegRounds rounds of one edge per 16-lane group (16 rounds measured 28 us at q = 100 000
// against 19 for the thread-per-edge form: too few workgroups, each a serial chain; 4 rounds: 1 563 workgroups)
constexpr int kRegRounds = 4;
constexpr int kRegEdges = kRegRounds * (kT / 16);
// per block partials: [0]=sum bce, [1]=sum (w-cos)^2, [2]=#valid, [3]=sum labels
__global__ void __launch_bounds__(kT) reg_fwd_partial(const float* __restrict__ w, const int64_t* __restrict__ sei, int64_t q,
                                                     const float* __restrict__ logits, int64_t C, const int64_t* __restrict__ y,
                                                     const uint8_t* __restrict__ tm, float* __restrict__ cos_out,
                                                     float* __restrict__ part) {
    __shared__ float red[kT / 64];
    // 16 lanes per edge (a logits row is C = 41 floats: three 64-byte segments instead of 41 strided scalar loads by one thread), the
    // workgroup's kRegEdges edges in kRegRounds rounds of kT / 16; lane 0 of a group keeps the group's terms
    const int sub = threadIdx.x & 15, grp = threadIdx.x >> 4;
    float bce = 0.f, sq = 0.f, nv = 0.f, nl = 0.f;
#pragma unroll
    for (int r = 0; r < kRegRounds; ++r) {
        const int64_t j = static_cast<int64_t>(blockIdx.x) * kRegEdges + r * (kT / 16) + grp;
        const bool live = j < q;
        const int64_t jj = live ? j : 0;
        const int64_t s = sei[jj], d = sei[q + jj];
        const float* x = logits + s * C;
        const float* yv = logits + d * C;
        float dot = 0.f, nx = 0.f, ny = 0.f;
        for (int64_t c = sub; c < C; c += 16) {
            const float a = x[c], b = yv[c];
            dot = fmaf(a, b, dot); nx = fmaf(a, a, nx); ny = fmaf(b, b, ny);
        }
        dot = row16_sum_all_dpp(dot); nx = row16_sum_all_dpp(nx); ny = row16_sum_all_dpp(ny);
        if (live && sub == 0) {
            const float wj = w[j];
            const float cs = cos_from(EdgeTerms{dot, nx, ny});
            if (cos_out) cos_out[j] = cs;
            const float df = wj - cs;
            sq += df * df;
            if (tm[s] && tm[d]) {
                nv += 1.f;
                const bool same = y[s] == y[d];
                nl += same ? 1.f : 0.f;
                // F.binary_cross_entropy clamps each log at -100
                bce += same ? -fmaxf(logf(wj), -100.f) : -fmaxf(logf(1.f - wj), -100.f);
            }
        }
    }
    const float r0 = block_sum(bce, red);
    const float r1 = block_sum(sq, red);
    const float r2 = block_sum(nv, red);
    const float r3 = block_sum(nl, red);
    if (threadIdx.x == 0) {
        part[4 * blockIdx.x + 0] = r0; part[4 * blockIdx.x + 1] = r1;
        part[4 * blockIdx.x + 2] = r2; part[4 * blockIdx.x + 3] = r3;
    }
}

// out[0] = reg1 (0 unless sum(labels) > 1), out[1] = reg2, out[2] = #valid, out[3] = sum labels,
// out[4] = coef1 * reg1 + coef2 * reg2
__global__ void __launch_bounds__(kT) reg_fwd_final(const float* __restrict__ part, int64_t nblk, int64_t q, float coef1, float coef2,
                                                   float* __restrict__ out) {
    __shared__ float red[kT / 64];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int64_t b = threadIdx.x; b < nblk; b += kT) {
        a0 += part[4 * b]; a1 += part[4 * b + 1]; a2 += part[4 * b + 2]; a3 += part[4 * b + 3];
    }
    const float r0 = block_sum(a0, red), r1 = block_sum(a1, red), r2 = block_sum(a2, red), r3 = block_sum(a3, red);
    if (threadIdx.x == 0) {
        const float reg1 = (r3 > 1.f) ? r0 / r2 : 0.f;             // training_hybrid.py:125-128
        const float reg2 = r1 / static_cast<float>(q);
        out[0] = reg1; out[1] = reg2; out[2] = r2; out[3] = r3;
        out[4] = coef1 * reg1 + coef2 * reg2;
    }
}

// The learned branch's whole loss in one finishing launch (training_hybrid.py:105-133): the cross entropy's mean over the train rows
// (as ce_final) and the two regularisers (as reg_fwd_final);  out[0..4] as reg_fwd_final, out[5] = cross entropy,
// out[6] = out[5] + out[4] = the loss.
__global__ void __launch_bounds__(kT) hybrid_loss_final(const float* __restrict__ part, int64_t nblk, int64_t q, float coef1, float coef2,
                                                       const float* __restrict__ rowloss, const uint8_t* __restrict__ mask, int64_t N,
                                                       float* __restrict__ out, int* __restrict__ n_rows) {
    __shared__ float red[kT / 64];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, ce = 0.f, cnt = 0.f;
    for (int64_t b = threadIdx.x; b < nblk; b += kT) {
        a0 += part[4 * b]; a1 += part[4 * b + 1]; a2 += part[4 * b + 2]; a3 += part[4 * b + 3];
    }
    for (int64_t i = threadIdx.x; i < N; i += kT) { ce += rowloss[i]; cnt += mask[i] ? 1.f : 0.f; }     // (counts < 2^24: exact in fp32)
    const float r0 = block_sum(a0, red), r1 = block_sum(a1, red), r2 = block_sum(a2, red), r3 = block_sum(a3, red);
    const float rc = block_sum(ce, red), rn = block_sum(cnt, red);
    if (threadIdx.x == 0) {
        const float reg1 = (r3 > 1.f) ? r0 / r2 : 0.f;
        const float reg2 = r1 / static_cast<float>(q);
        out[0] = reg1; out[1] = reg2; out[2] = r2; out[3] = r3;
        out[4] = coef1 * reg1 + coef2 * reg2;
        out[5] = rc / rn;                                  // 0/0 = nan, as torch's mean over an empty selection
        out[6] = out[5] + out[4];
        n_rows[0] = static_cast<int>(rn);
    }
}

// ce_bwd on top of a gradient that is already there: dlogits += (softmax - onehot) g / #train on the train rows.
__global__ void __launch_bounds__(kT) ce_bwd_acc(const float* __restrict__ logits, int64_t N, int64_t C, const int64_t* __restrict__ y,
                                                const uint8_t* __restrict__ mask, const float* __restrict__ row_lse,
                                                const int* __restrict__ n_rows, const float* __restrict__ grad_loss,
                                                float* __restrict__ dlogits) {
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x;
    if (idx >= N * C) return;
    const int64_t i = idx / C, c = idx - i * C;
    if (!mask[i]) return;
    const float sm = expf(logits[idx] - row_lse[i]);
    dlogits[idx] += (sm - (y[i] == c ? 1.f : 0.f)) * (grad_loss[0] / static_cast<float>(n_rows[0]));
}

// raw[0..3] = {sum bce, sum (w-cos)^2, #valid, sum labels} of this rank's edges (edge-sharded losses: the
// ranks all-reduce these four sums and finish the formulas with the global q).
__global__ void __launch_bounds__(kT) reg_raw_final(const float* __restrict__ part, int64_t nblk, float* __restrict__ raw) {
    __shared__ float red[kT / 64];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int64_t b = threadIdx.x; b < nblk; b += kT) {
        a0 += part[4 * b]; a1 += part[4 * b + 1]; a2 += part[4 * b + 2]; a3 += part[4 * b + 3];
    }
    const float r0 = block_sum(a0, red), r1 = block_sum(a1, red), r2 = block_sum(a2, red), r3 = block_sum(a3, red);
    if (threadIdx.x == 0) { raw[0] = r0; raw[1] = r1; raw[2] = r2; raw[3] = r3; }
}

// Per sampled edge: dw[j] and the gradient rows wrt logits[src] (Gs) and logits[dst] (Gd).
// One 16-lane group per sampled edge (16 edges per workgroup): the group's lanes stride the C logit columns, so the two
// [q, C] gradient-row arrays are written as contiguous segments (a thread-per-edge loop writes them with a stride of C
// floats: 69 us at q = 100 000, C = 41).  The three dot products are reduced inside the group in a fixed order.
__global__ void __launch_bounds__(kT) reg_bwd_edges(const float* __restrict__ w, const int64_t* __restrict__ sei, int64_t q,
                                                   const float* __restrict__ logits, int64_t C, const int64_t* __restrict__ y,
                                                   const uint8_t* __restrict__ tm, const float* __restrict__ out, float coef1,
                                                   float coef2, float q_norm, const float* __restrict__ grad_loss,
                                                   float* __restrict__ dw, float* __restrict__ Gs, float* __restrict__ Gd) {
    const int sub = threadIdx.x & 15;
    const int64_t j = static_cast<int64_t>(blockIdx.x) * (kT / 16) + (threadIdx.x >> 4);
    const bool live = j < q;
    const int64_t jj = live ? j : 0;
    const float gl = grad_loss[0];
    const int64_t s = sei[jj], d = sei[q + jj];
    const float* x = logits + s * C;
    const float* yv = logits + d * C;
    float dot = 0.f, nx = 0.f, ny = 0.f;
    for (int64_t c = sub; c < C; c += 16) {
        const float a = x[c], b = yv[c];
        dot = fmaf(a, b, dot); nx = fmaf(a, a, nx); ny = fmaf(b, b, ny);
    }
    dot = row16_sum_all_dpp(dot); nx = row16_sum_all_dpp(nx); ny = row16_sum_all_dpp(ny);      // all 16 lanes of a group end with the same sums
    if (!live) return;
    const float den2 = fmaxf(nx * ny, 1e-16f);
    const float inv = 1.0f / sqrtf(den2);
    const float cs = dot * inv;
    const float wj = w[j];
    const float r = 2.0f * (wj - cs) / q_norm * coef2 * gl;                        // dL/d(w - cos); q_norm = global #sampled edges
    if (sub == 0) {
        float g = r;
        if (coef1 != 0.f && out[3] > 1.f && tm[s] && tm[d]) {
            const float tgt = (y[s] == y[d]) ? 1.f : 0.f;
            g += coef1 * gl * (wj - tgt) / fmaxf((1.f - wj) * wj, 1e-12f) / out[2];   // torch's BCE backward
        }
        dw[j] = g;
    }
    // d cos / d x = y * inv - cos * x / |x|^2  (only while the eps clamp is inactive, as autograd)
    const bool clamped = nx * ny < 1e-16f;
    const float cx = clamped ? 0.f : cs / nx, cy = clamped ? 0.f : cs / ny;
    for (int64_t c = sub; c < C; c += 16) {
        const float a = x[c], b = yv[c];
        Gs[j * C + c] = -r * (b * inv - cx * a);
        Gd[j * C + c] = -r * (a * inv - cy * b);
    }
}

}  // namespace
}  // namespace sgs

using namespace sgs;

extern "C" {

int sgs_masked_correct(const float* logits, int64_t N, int64_t C, const int64_t* y, const uint8_t* train_mask,
                       int32_t* correct, sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(N >= 0 && C > 0 && correct, SGS_EINVAL, "sgs_masked_correct: bad arguments");
    if (int rc = zero_async(correct, 8, stream)) return rc;
    if (N == 0) return SGS_OK;
    SGS_REQUIRE(logits && y && train_mask, SGS_EINVAL, "sgs_masked_correct: null pointer");
    hipLaunchKernelGGL(masked_correct, dim3(cdiv(N * 64, kT)), dim3(kT), 0, stream, logits, N, C, y, train_mask, correct);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

int sgs_masked_correct_pair(const float* logits_a, const float* logits_b, int64_t N, int64_t C, const int64_t* y,
                            const uint8_t* train_mask, int32_t* correct4, sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(N >= 0 && C > 0 && correct4, SGS_EINVAL, "sgs_masked_correct_pair: bad arguments");
    if (N == 0) return SGS_OK;
    SGS_REQUIRE(logits_a && logits_b && y && train_mask, SGS_EINVAL, "sgs_masked_correct_pair: null pointer");
    hipLaunchKernelGGL(masked_correct_pair, dim3(static_cast<unsigned>(cdiv(N, 16)), 2), dim3(1024), 0, stream, logits_a, logits_b, N, C, y, train_mask,
                       correct4);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

size_t sgs_gate_counts_workspace_bytes(int64_t N) { return carve_bytes(static_cast<size_t>(2 * (cdiv(N < 0 ? 0 : N, 16) + 1)), 8) + 256; }

int sgs_gate_counts(const float* logits_a, const float* logits_b, int64_t N, int64_t C, const int64_t* y, const uint8_t* train_mask,
                    int32_t* out5, const uint64_t* seq_dev, int32_t* dst_host_mapped, void* ws, size_t ws_bytes, sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(N >= 0 && N < (int64_t(1) << 30) && C > 0 && out5, SGS_EINVAL, "sgs_gate_counts: bad arguments");
    SGS_REQUIRE(N == 0 || (logits_a && logits_b && y && train_mask), SGS_EINVAL, "sgs_gate_counts: null pointer");
    SGS_REQUIRE(ws && ws_bytes >= sgs_gate_counts_workspace_bytes(N), SGS_EWORKSPACE, "sgs_gate_counts: workspace too small");
    Carver cv(ws);
    const int nblk = static_cast<int>(cdiv(N, 16));
    int2* part = reinterpret_cast<int2*>(cv.take<int64_t>(static_cast<size_t>(2 * (nblk + 1))));
    if (nblk > 0)
        hipLaunchKernelGGL(gate_counts_partial, dim3(static_cast<unsigned>(nblk), 2), dim3(1024), 0, stream, logits_a, logits_b, N, C, y, train_mask,
                           part);
    hipLaunchKernelGGL(gate_counts_finish, dim3(1), dim3(64), 0, stream, static_cast<const int2*>(part), nblk, out5, seq_dev, dst_host_mapped);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

int sgs_loss_tick(float* loss_sum, const float* loss, uint64_t* epoch, sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    hipLaunchKernelGGL(loss_tick, dim3(1), dim3(64), 0, stream, loss_sum, loss, epoch);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

int sgs_publish_to_host(const int32_t* src_dev, int64_t n, const uint64_t* seq_dev, int32_t* dst_host_mapped, sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(n > 0 && n <= 63 && src_dev && dst_host_mapped, SGS_EINVAL, "sgs_publish_to_host: bad arguments");
    hipLaunchKernelGGL(publish_words, dim3(1), dim3(64), 0, stream, src_dev, static_cast<int>(n), seq_dev, dst_host_mapped);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

int sgs_masked_ce_fwd(const float* logits, int64_t N, int64_t C, const int64_t* y, const uint8_t* train_mask, float* loss,
                      float* row_lse, float* rowloss, int32_t* n_rows, sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(N > 0 && C > 0 && logits && y && train_mask && loss && row_lse && rowloss && n_rows, SGS_EINVAL,
                "sgs_masked_ce_fwd: bad arguments");
    hipLaunchKernelGGL(ce_rows, dim3(cdiv(N * 64, kT)), dim3(kT), 0, stream, logits, N, C, y, train_mask, row_lse, rowloss);
    hipLaunchKernelGGL(ce_final, dim3(1), dim3(1024), 0, stream, rowloss, train_mask, N, loss, n_rows);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

int sgs_masked_ce_bwd(const float* logits, int64_t N, int64_t C, const int64_t* y, const uint8_t* train_mask,
                      const float* row_lse, const int32_t* n_rows, const float* grad_loss, float* dlogits,
                      sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(N > 0 && C > 0 && logits && y && train_mask && row_lse && n_rows && grad_loss && dlogits, SGS_EINVAL,
                "sgs_masked_ce_bwd: bad arguments");
    hipLaunchKernelGGL(ce_bwd, dim3(cdiv(N * C, kT)), dim3(kT), 0, stream, logits, N, C, y, train_mask, row_lse, n_rows, grad_loss,
                       dlogits);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

size_t sgs_edge_reg_workspace_bytes(int64_t q) { return carve_bytes(4 * (cdiv(q < 0 ? 0 : q, kRegEdges) + 1), 4) + 256; }

int sgs_edge_reg_fwd(const float* w, const int64_t* sampled_edge_index, int64_t q, const float* logits, int64_t N, int64_t C,
                     const int64_t* y, const uint8_t* train_mask, float coef1, float coef2, float* out, float* cos_out,
                     void* ws, size_t ws_bytes, sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(q > 0 && N > 0 && C > 0 && w && sampled_edge_index && logits && y && train_mask && out, SGS_EINVAL,
                "sgs_edge_reg_fwd: bad arguments");
    SGS_REQUIRE(ws && ws_bytes >= sgs_edge_reg_workspace_bytes(q), SGS_EWORKSPACE, "sgs_edge_reg_fwd: workspace too small");
    Carver cv(ws);
    const int64_t nblk = cdiv(q, kRegEdges);
    float* part = cv.take<float>(4 * (nblk + 1));
    hipLaunchKernelGGL(reg_fwd_partial, dim3(nblk), dim3(kT), 0, stream, w, sampled_edge_index, q, logits, C, y, train_mask, cos_out,
                       part);
    hipLaunchKernelGGL(reg_fwd_final, dim3(1), dim3(kT), 0, stream, part, nblk, q, coef1, coef2, out);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

int sgs_hybrid_loss_fwd(const float* logits, int64_t N, int64_t C, const int64_t* y, const uint8_t* train_mask, const float* w,
                        const int64_t* sampled_edge_index, int64_t q, float coef1, float coef2, float* out, float* row_lse, float* rowloss,
                        int32_t* n_rows, void* ws, size_t ws_bytes, sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(q > 0 && N > 0 && N < (int64_t(1) << 24) && C > 0 && logits && y && train_mask && w && sampled_edge_index && out && row_lse &&
                    rowloss && n_rows,
                SGS_EINVAL, "sgs_hybrid_loss_fwd: bad arguments");
    SGS_REQUIRE(ws && ws_bytes >= sgs_edge_reg_workspace_bytes(q), SGS_EWORKSPACE, "sgs_hybrid_loss_fwd: workspace too small");
    Carver cv(ws);
    const int64_t nblk = cdiv(q, kRegEdges);
    float* part = cv.take<float>(4 * (nblk + 1));
    hipLaunchKernelGGL(ce_rows, dim3(cdiv(N * 64, kT)), dim3(kT), 0, stream, logits, N, C, y, train_mask, row_lse, rowloss);
    hipLaunchKernelGGL(reg_fwd_partial, dim3(nblk), dim3(kT), 0, stream, w, sampled_edge_index, q, logits, C, y, train_mask,
                       static_cast<float*>(nullptr), part);
    hipLaunchKernelGGL(hybrid_loss_final, dim3(1), dim3(kT), 0, stream, part, nblk, q, coef1, coef2, rowloss, train_mask, N, out, n_rows);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

int sgs_masked_ce_bwd_acc(const float* logits, int64_t N, int64_t C, const int64_t* y, const uint8_t* train_mask, const float* row_lse,
                          const int32_t* n_rows, const float* grad_loss, float* dlogits, sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(N > 0 && C > 0 && logits && y && train_mask && row_lse && n_rows && grad_loss && dlogits, SGS_EINVAL,
                "sgs_masked_ce_bwd_acc: bad arguments");
    hipLaunchKernelGGL(ce_bwd_acc, dim3(cdiv(N * C, kT)), dim3(kT), 0, stream, logits, N, C, y, train_mask, row_lse, n_rows, grad_loss,
                       dlogits);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

int sgs_edge_reg_partial(const float* w, const int64_t* sampled_edge_index, int64_t q, const float* logits, int64_t N, int64_t C,
                         const int64_t* y, const uint8_t* train_mask, float* raw, void* ws, size_t ws_bytes, sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(q >= 0 && N > 0 && C > 0 && raw, SGS_EINVAL, "sgs_edge_reg_partial: bad arguments");
    SGS_REQUIRE(ws && ws_bytes >= sgs_edge_reg_workspace_bytes(q), SGS_EWORKSPACE, "sgs_edge_reg_partial: workspace too small");
    Carver cv(ws);
    const int64_t nblk = cdiv(q, kRegEdges);
    float* part = cv.take<float>(4 * (nblk + 1));
    if (q > 0) {
        SGS_REQUIRE(w && sampled_edge_index && logits && y && train_mask, SGS_EINVAL, "sgs_edge_reg_partial: null pointer");
        hipLaunchKernelGGL(reg_fwd_partial, dim3(nblk), dim3(kT), 0, stream, w, sampled_edge_index, q, logits, C, y, train_mask,
                           static_cast<float*>(nullptr), part);
    }
    hipLaunchKernelGGL(reg_raw_final, dim3(1), dim3(kT), 0, stream, part, nblk, raw);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

int sgs_edge_reg_bwd(const float* w, const int64_t* sampled_edge_index, int64_t q, int64_t q_global, const float* logits, int64_t N,
                     int64_t C, const int64_t* y, const uint8_t* train_mask, const float* out, float coef1, float coef2,
                     const float* grad_loss, float* dw, float* Gs, float* Gd, sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(q >= 0 && q_global >= q && N > 0 && C > 0, SGS_EINVAL, "sgs_edge_reg_bwd: bad arguments");
    if (q == 0) return SGS_OK;
    SGS_REQUIRE(w && sampled_edge_index && logits && y && train_mask && out && grad_loss && dw && Gs && Gd, SGS_EINVAL,
                "sgs_edge_reg_bwd: null pointer");
    hipLaunchKernelGGL(reg_bwd_edges, dim3(cdiv(q, kT / 16)), dim3(kT), 0, stream, w, sampled_edge_index, q, logits, C, y, train_mask, out,
                       coef1, coef2, static_cast<float>(q_global), grad_loss, dw, Gs, Gd);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

}  // extern "C"
