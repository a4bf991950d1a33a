// Adam step over a whole parameter group in ONE launch (torch.optim.Adam's update rule: coupled weight decay, no
// amsgrad).  Why here: the partition-scale training step is launch-latency bound, the reference steps two Adam
// optimisers per batch (training_hybrid.py:22-27, 135-141), and torch's fused/foreach kernels hand each workgroup a
// 64 Ki-element chunk -- three workgroups for the ~165 k parameters of the GCN group, 38 us per step on MI355X.  This
// kernel maps 2048-element chunks of all tensors onto the grid; the tensor descriptors travel BY VALUE in the kernel
// arguments (no device-side table to build or keep alive, and a captured HIP graph owns its copy), and the per-parameter
// step counters live on the device, so the step is capturable.
#include "sgs_common.h"

namespace sgs {
namespace {

constexpr int kAT = 256;
constexpr int kAChunk = 2048;
constexpr int kAMax = 24;           // tensors per launch

struct AdamDesc {                   // 7 x 64-bit words per tensor, same layout as the caller's int64 array
    float* p;
    const float* g;
    float* m;
    float* v;
    int64_t n;
    float* step;                    // this tensor's completed steps (torch keeps one counter per parameter)
    const float* gate;              // optional device predicate: the tensor is left untouched while *gate == 0
};
struct AdamArgs {
    AdamDesc d[kAMax];
    int chunk_end[kAMax];           // exclusive prefix end of each tensor's chunks in the grid
    int n_tensors;
    float lr, beta1, beta2, eps, weight_decay;
    int maximize;
    unsigned int* ticket;
};

__global__ void __launch_bounds__(kAT) adam_group(AdamArgs a) {
    __shared__ unsigned int s_last;
    int ti = 0;
    while (static_cast<int>(blockIdx.x) >= a.chunk_end[ti]) ++ti;      // uniform; <= kAMax steps
    const int chunk = static_cast<int>(blockIdx.x) - (ti ? a.chunk_end[ti - 1] : 0);
    const AdamDesc d = a.d[ti];
    const bool open = d.gate == nullptr || d.gate[0] != 0.f;   // data-parallel runs: "no rank took the learned branch" skips the scorer's tensors
    const float t = d.step[0] + 1.0f;                 // read by every workgroup before any of them can bump it (below)
    // bias corrections in double: beta^t for t up to millions
    const double bc1 = 1.0 - exp(static_cast<double>(t) * log(static_cast<double>(a.beta1)));
    const double bc2 = 1.0 - exp(static_cast<double>(t) * log(static_cast<double>(a.beta2)));
    const float step_size = static_cast<float>(static_cast<double>(a.lr) / bc1);
    const float inv_sqrt_bc2 = static_cast<float>(1.0 / sqrt(bc2));
    const float beta1 = a.beta1, beta2 = a.beta2, eps = a.eps, wd = a.weight_decay;
    const int64_t base = static_cast<int64_t>(chunk) * kAChunk;
#pragma unroll
    for (int it = 0; it < kAChunk / kAT; ++it) {
        const int64_t i = base + static_cast<int64_t>(it) * kAT + threadIdx.x;
        if (open && i < d.n) {
            float g = d.g[i];
            if (a.maximize) g = -g;
            const float p = d.p[i];
            if (wd != 0.f) g = fmaf(wd, p, g);
            float m = d.m[i], v = d.v[i];
            m = m + (g - m) * (1.0f - beta1);                         // lerp, as torch
            v = beta2 * v + (1.0f - beta2) * g * g;
            const float denom = sqrtf(v) * inv_sqrt_bc2 + eps;
            d.m[i] = m;
            d.v[i] = v;
            d.p[i] = p - step_size * (m / denom);
        }
    }
    // The workgroup that arrives last bumps the counters: every other workgroup has read its `step` before arriving
    // (its update depends on that read), so no fence is needed -- device-scope fences are expensive on this part.
    __syncthreads();
    if (threadIdx.x == 0) s_last = (atomicAdd(a.ticket, 1u) == gridDim.x - 1) ? 1u : 0u;
    __syncthreads();
    if (s_last) {
        if (static_cast<int>(threadIdx.x) < a.n_tensors) {
            const AdamDesc& dd = a.d[threadIdx.x];
            if (dd.gate == nullptr || dd.gate[0] != 0.f) dd.step[0] += 1.0f;
        }
        if (threadIdx.x == 0) __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---------------------------------------------------------------------------------------------
// Both optimisers of a learned step in ONE launch (training_hybrid.py:136-137: optimizer_edge_prob.step(); optimizer_gnn.step()), with
// the step's closing bookkeeping (sgs_loss_tick) done by the last workgroup.  The reference's two Adam objects OVERLAP on
// edge_prob_mlp.gcn* (main.py:100-109, 122), which therefore step twice per learned batch with the same gradient: a descriptor carries
// an optional SECOND state, applied to the same elements right after the first while they are still in registers -- exactly the
// sequence of two launches, three fewer dependent launches per step.
constexpr int kAMax2 = 16;
struct AdamState {
    float* m;
    float* v;
    float* step;
    const float* gate;
    float lr, beta1, beta2, eps, wd;
    int maximize;
};
struct AdamDesc2 {
    float* p;
    const float* g;
    int64_t n;
    AdamState s[2];
    int n_states;
};
struct AdamArgs2 {
    AdamDesc2 d[kAMax2];
    int chunk_end[kAMax2];
    int n_tensors;
    unsigned int* ticket;
    float* loss_sum;
    const float* loss;
    unsigned long long* epoch;
};

__global__ void __launch_bounds__(kAT) adam_multi(AdamArgs2 a) {
    __shared__ unsigned int s_last;
    int ti = 0;
    while (static_cast<int>(blockIdx.x) >= a.chunk_end[ti]) ++ti;
    const int chunk = static_cast<int>(blockIdx.x) - (ti ? a.chunk_end[ti - 1] : 0);
    const AdamDesc2& d = a.d[ti];
    bool open[2];
    float step_size[2], inv_sqrt_bc2[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        open[q] = q < d.n_states && (d.s[q].gate == nullptr || d.s[q].gate[0] != 0.f);
        step_size[q] = 0.f;
        inv_sqrt_bc2[q] = 0.f;
        if (q < d.n_states) {
            const float t = d.s[q].step[0] + 1.0f;
            const double bc1 = 1.0 - exp(static_cast<double>(t) * log(static_cast<double>(d.s[q].beta1)));
            const double bc2 = 1.0 - exp(static_cast<double>(t) * log(static_cast<double>(d.s[q].beta2)));
            step_size[q] = static_cast<float>(static_cast<double>(d.s[q].lr) / bc1);
            inv_sqrt_bc2[q] = static_cast<float>(1.0 / sqrt(bc2));
        }
    }
    const int64_t base = static_cast<int64_t>(chunk) * kAChunk;
#pragma unroll
    for (int it = 0; it < kAChunk / kAT; ++it) {
        const int64_t i = base + static_cast<int64_t>(it) * kAT + threadIdx.x;
        if (i < d.n && (open[0] || open[1])) {
            const float g0 = d.g[i];
            float p = d.p[i];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (open[q]) {
                    const AdamState& st = d.s[q];
                    float g = st.maximize ? -g0 : g0;
                    if (st.wd != 0.f) g = fmaf(st.wd, p, g);
                    float m = st.m[i], v = st.v[i];
                    m = m + (g - m) * (1.0f - st.beta1);
                    v = st.beta2 * v + (1.0f - st.beta2) * g * g;
                    const float denom = sqrtf(v) * inv_sqrt_bc2[q] + st.eps;
                    st.m[i] = m;
                    st.v[i] = v;
                    p = p - step_size[q] * (m / denom);
                }
            }
            d.p[i] = p;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) s_last = (atomicAdd(a.ticket, 1u) == gridDim.x - 1) ? 1u : 0u;
    __syncthreads();
    if (s_last) {
        if (static_cast<int>(threadIdx.x) < 2 * a.n_tensors) {
            const AdamDesc2& dd = a.d[threadIdx.x >> 1];
            const int q = threadIdx.x & 1;
            if (q < dd.n_states && (dd.s[q].gate == nullptr || dd.s[q].gate[0] != 0.f)) {
                // two states of ONE optimiser never share a counter; the same tensor's two states belong to two optimisers
                dd.s[q].step[0] += 1.0f;
            }
        }
        if (threadIdx.x == 64) {                                       // the step's closing bookkeeping (sgs_loss_tick)
            if (a.loss_sum && a.loss) a.loss_sum[0] += a.loss[0];
            if (a.epoch) a.epoch[0] += 1ull;
        }
        if (threadIdx.x == 0) __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace
}  // namespace sgs

using namespace sgs;

extern "C" {

int sgs_adam_max_tensors(void) { return kAMax; }

int sgs_adam_step(const int64_t* desc_host, int64_t n_tensors, float lr, float beta1, float beta2, float eps, float weight_decay,
                  int maximize, uint32_t* ticket, sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(n_tensors >= 0 && n_tensors <= kAMax, SGS_EINVAL, "sgs_adam_step: at most %d tensors per call", kAMax);
    if (n_tensors == 0) return SGS_OK;
    SGS_REQUIRE(desc_host && ticket, SGS_EINVAL, "sgs_adam_step: null pointer");
    SGS_REQUIRE(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f, SGS_EINVAL, "sgs_adam_step: bad hyper-parameters");
    AdamArgs a{};
    int64_t total = 0;
    for (int i = 0; i < kAMax; ++i) {
        if (i < n_tensors) {
            const int64_t* w = desc_host + 7 * i;
            a.d[i].p = reinterpret_cast<float*>(w[0]);
            a.d[i].g = reinterpret_cast<const float*>(w[1]);
            a.d[i].m = reinterpret_cast<float*>(w[2]);
            a.d[i].v = reinterpret_cast<float*>(w[3]);
            a.d[i].n = w[4];
            a.d[i].step = reinterpret_cast<float*>(w[5]);
            a.d[i].gate = reinterpret_cast<const float*>(w[6]);
            SGS_REQUIRE(a.d[i].p && a.d[i].g && a.d[i].m && a.d[i].v && a.d[i].step && a.d[i].n >= 0, SGS_EINVAL,
                        "sgs_adam_step: bad descriptor %d", i);
            total += cdiv(a.d[i].n, kAChunk);
        }
        a.chunk_end[i] = static_cast<int>(total);
    }
    SGS_REQUIRE(total < (int64_t(1) << 31), SGS_EINVAL, "sgs_adam_step: too many chunks");
    if (total == 0) return SGS_OK;
    a.n_tensors = static_cast<int>(n_tensors);
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay; a.maximize = maximize;
    a.ticket = ticket;
    hipLaunchKernelGGL(adam_group, dim3(static_cast<unsigned>(total)), dim3(kAT), 0, stream, a);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

int sgs_adam_multi_max_tensors(void) { return kAMax2; }

int sgs_adam_step_multi(const int64_t* desc_host, const float* hyper_host, int64_t n_tensors, uint32_t* ticket, float* loss_sum, const float* loss,
                        uint64_t* epoch, sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(n_tensors >= 0 && n_tensors <= kAMax2, SGS_EINVAL, "sgs_adam_step_multi: at most %d tensors per call", kAMax2);
    SGS_REQUIRE(ticket && (n_tensors == 0 || (desc_host && hyper_host)), SGS_EINVAL, "sgs_adam_step_multi: null pointer");
    AdamArgs2 a{};
    int64_t total = 0;
    for (int i = 0; i < kAMax2; ++i) {
        if (i < n_tensors) {
            const int64_t* w = desc_host + 11 * i;       // {param, grad, numel, m, v, step, gate, m2, v2, step2, gate2}
            const float* h = hyper_host + 12 * i;        // {lr, beta1, beta2, eps, weight_decay, maximize} x 2
            AdamDesc2& d = a.d[i];
            d.p = reinterpret_cast<float*>(w[0]);
            d.g = reinterpret_cast<const float*>(w[1]);
            d.n = w[2];
            d.n_states = w[7] ? 2 : 1;
            for (int q = 0; q < d.n_states; ++q) {
                AdamState& st = d.s[q];
                st.m = reinterpret_cast<float*>(w[3 + 4 * q]);
                st.v = reinterpret_cast<float*>(w[4 + 4 * q]);
                st.step = reinterpret_cast<float*>(w[5 + 4 * q]);
                st.gate = reinterpret_cast<const float*>(w[6 + 4 * q]);
                st.lr = h[6 * q]; st.beta1 = h[6 * q + 1]; st.beta2 = h[6 * q + 2]; st.eps = h[6 * q + 3]; st.wd = h[6 * q + 4];
                st.maximize = h[6 * q + 5] != 0.f;
                SGS_REQUIRE(st.m && st.v && st.step && st.beta1 >= 0.f && st.beta1 < 1.f && st.beta2 >= 0.f && st.beta2 < 1.f && st.eps >= 0.f, SGS_EINVAL,
                            "sgs_adam_step_multi: bad state %d of descriptor %d", q, i);
            }
            SGS_REQUIRE(d.p && d.g && d.n >= 0, SGS_EINVAL, "sgs_adam_step_multi: bad descriptor %d", i);
            total += cdiv(d.n, kAChunk);
        }
        a.chunk_end[i] = static_cast<int>(total);
    }
    SGS_REQUIRE(total < (int64_t(1) << 31), SGS_EINVAL, "sgs_adam_step_multi: too many chunks");
    a.n_tensors = static_cast<int>(n_tensors);
    a.ticket = ticket;
    a.loss_sum = loss_sum; a.loss = loss; a.epoch = reinterpret_cast<unsigned long long*>(epoch);
    if (total == 0) {
        if ((loss_sum && loss) || epoch) return sgs_loss_tick(loss_sum, loss, epoch, stream_);
        return SGS_OK;
    }
    hipLaunchKernelGGL(adam_multi, dim3(static_cast<unsigned>(total)), dim3(kAT), 0, stream, a);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

}  // extern "C"
