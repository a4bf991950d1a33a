// Effective-resistance edge prior: the random-walk estimator the reference runs as a Python /
// networkx loop over every edge (EffectiveResistanceWeights.ipynb cell 11 `er_edge`, used by datasets.py:159-173 add_ER):
//   for walk length i = 0 .. l-1:   r walks from s and r walks from t;  X_is, X_it = #walks from s ending at s / t,  Y_is, Y_it likewise from t
//     delta += ( X_is/deg(s) - X_it/deg(t) - Y_is/deg(s) + Y_it/deg(t) ) / r
//   weight(s,t) = max(0, delta)                                   (l = 4, r = 100 in the reference)
// One wave per edge; lane j runs walks j, j+64, ... of both endpoints; every step is a uniform choice among the
// neighbours of the current node read from the CSR of the (symmetric, coalesced) edge list -- which is what
// to_networkx(..., to_undirected=True) walks on.  Counter-based randomness (Philox, keyed on seed, edge, walk, step), so
// the result does not depend on scheduling.  Latency-bound random gathers; embarrassingly parallel.
#include "sgs_common.h"

namespace sgs {
namespace {

__device__ __forceinline__ int walk(const int* __restrict__ ptr, const int* __restrict__ nbr, int v, int len, uint64_t seed,
                                    uint64_t edge, uint32_t walk_id) {
    for (int step = 0; step < len; ++step) {
        const int b = ptr[v], d = ptr[v + 1] - b;
        if (d == 0) continue;                                   // isolated node: stays (as the reference's `continue`)
        const Philox4 r = philox4x32_10(static_cast<uint32_t>(edge), static_cast<uint32_t>(edge >> 32), walk_id, static_cast<uint32_t>(step),
                                        static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32));
        v = nbr[b + static_cast<int>(mulhi32(r.v[0], static_cast<uint32_t>(d)))];       // uniform in [0, d)
    }
    return v;
}

__global__ void __launch_bounds__(256) er_weight_kernel(const int64_t* __restrict__ ei, int64_t E, const int* __restrict__ ptr,
                                                       const int* __restrict__ nbr, int l, int r, uint64_t seed,
                                                       float* __restrict__ weight) {
    const int lane = threadIdx.x & 63;
    const int64_t e = (static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x) >> 6;
    if (e >= E) return;
    const int s = static_cast<int>(ei[e]), t = static_cast<int>(ei[E + e]);
    const float ds = static_cast<float>(ptr[s + 1] - ptr[s]), dt = static_cast<float>(ptr[t + 1] - ptr[t]);
    float delta = 0.f;
    for (int i = 0; i < l; ++i) {
        int xis = 0, xit = 0, yis = 0, yit = 0;
        for (int j = lane; j < r; j += 64) {
            const uint32_t wid = static_cast<uint32_t>((i * r + j) * 2);
            const int v = walk(ptr, nbr, s, i, seed, static_cast<uint64_t>(e), wid);
            xis += v == s; xit += v == t;
            const int u = walk(ptr, nbr, t, i, seed, static_cast<uint64_t>(e), wid + 1);
            yis += u == s; yit += u == t;
        }
        xis = wave_sum_int_all(xis); xit = wave_sum_int_all(xit); yis = wave_sum_int_all(yis); yit = wave_sum_int_all(yit);
        const float di = static_cast<float>(xis) / ds - static_cast<float>(xit) / dt - static_cast<float>(yis) / ds + static_cast<float>(yit) / dt;
        delta += di / static_cast<float>(r);
    }
    if (lane == 0) weight[e] = fmaxf(0.f, delta);
}

}  // namespace
}  // namespace sgs

using namespace sgs;

extern "C" {

int sgs_er_weight(const int64_t* edge_index, int64_t E, int64_t N, const int32_t* out_ptr, const int32_t* out_dst, int walk_lengths,
                  int walks, uint64_t seed, float* weight, sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(E >= 0 && N >= 0 && walk_lengths >= 1 && walk_lengths <= 64 && walks >= 1 && walks <= (1 << 20), SGS_EINVAL,
                "sgs_er_weight: bad arguments");
    if (E == 0) return SGS_OK;
    SGS_REQUIRE(edge_index && out_ptr && out_dst && weight, SGS_EINVAL, "sgs_er_weight: null pointer");
    hipLaunchKernelGGL(er_weight_kernel, dim3(static_cast<unsigned>(cdiv(E * 64, 256))), dim3(256), 0, stream, edge_index, E, out_ptr, out_dst,
                       walk_lengths, walks, seed, weight);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

}  // extern "C"
