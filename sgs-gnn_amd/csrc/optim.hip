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

}  // extern "C"
