// K8: GAT attention (PyG 2.3.1 GATConv, heads = 1) over the dst-sorted CSR (gfx950).
//
// Reference: model.py:189-208 builds torch_geometric.nn.models.GAT(num_layers=2, act='relu',
// dropout=p); each GATConv layer is
//   x' = lin(x);  a_s = <x', att_src>, a_d = <x', att_dst>                  (node level, host side)
//   existing self loops removed, one loop per node added
//   e_k = leaky_relu(a_s[src_k] + a_d[dst_k], 0.2);  alpha = softmax over each node's in-edges
//   alpha = dropout(alpha, p)  (training);  out[i] = sum_k alpha_k x'[src_k] + bias
// `edge_weight` is ignored by PyG's GAT (supports_edge_weight = False).
// The aggregation itself reuses sgs_spmm_csr (val = alpha, diag = loop alpha); this file holds the
// segment softmax and its backward.  One wave per destination node; rows are walked three times
// (max, sum, normalise) from L2.
#include "sgs_common.h"

namespace sgs {
namespace {

constexpr int kT = 256;

__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.f ? v : slope * v; }

// alpha_in[k] (dst-CSR order; 0 for (i,i) entries), alpha_loop[i]: post-softmax, post-dropout weights;
// soft_in / soft_loop: the pre-dropout softmax values kept for backward.
__global__ void __launch_bounds__(kT) gat_alpha_fwd(const float* __restrict__ a_s, const float* __restrict__ a_d, int64_t N,
                                                   const int* __restrict__ in_ptr, const int* __restrict__ in_src,
                                                   const int* __restrict__ in_eid, float slope, float drop_scale,
                                                   uint32_t drop_thresh, int use_drop, uint64_t seed, uint32_t site,
                                                   const uint64_t* __restrict__ epoch, float* __restrict__ soft_in,
                                                   float* __restrict__ soft_loop, float* __restrict__ alpha_in,
                                                   float* __restrict__ alpha_loop) {
    seed = fold_epoch(seed, epoch);
    const int lane = threadIdx.x & 63;
    const int64_t i = (static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x) >> 6;
    if (i >= N) return;
    const int b = in_ptr[i], e = in_ptr[i + 1];
    const float ad = a_d[i];
    const float eloop = lrelu(a_s[i] + ad, slope);
    float mx = eloop;
    for (int k = b + lane; k < e; k += 64) {
        const int s = in_src[k];
        if (s != static_cast<int>(i)) mx = fmaxf(mx, lrelu(a_s[s] + ad, slope));
    }
    mx = wave_max_all(mx);
    float sum = 0.f;
    for (int k = b + lane; k < e; k += 64) {
        const int s = in_src[k];
        if (s != static_cast<int>(i)) sum += expf(lrelu(a_s[s] + ad, slope) - mx);
    }
    sum = wave_sum_all(sum) + expf(eloop - mx);
    const float inv = 1.0f / (sum + 1e-16f);                  // torch_geometric.utils.softmax: / (sum + 1e-16)
    // attention dropout is keyed by (site, row = edge id) for edges and (site + 1, row = node) for loops
    for (int k = b + lane; k < e; k += 64) {
        const int s = in_src[k];
        float sm = 0.f, al = 0.f;
        if (s != static_cast<int>(i)) {
            sm = expf(lrelu(a_s[s] + ad, slope) - mx) * inv;
            al = sm;
            if (use_drop) al = dropout_keep_at(seed, site, static_cast<uint64_t>(in_eid[k]), 0u, drop_thresh) ? sm * drop_scale : 0.f;
        }
        soft_in[k] = sm;
        alpha_in[k] = al;
    }
    if (lane == 0) {
        const float sm = expf(eloop - mx) * inv;
        float al = sm;
        if (use_drop) al = dropout_keep_at(seed, site + 1u, static_cast<uint64_t>(i), 0u, drop_thresh) ? sm * drop_scale : 0.f;
        soft_loop[i] = sm;
        alpha_loop[i] = al;
    }
}

// Backward of dropout + softmax + leaky_relu for one destination row:
//   dal_k  = galpha[eid_k] (gradient wrt the dropped alpha; from sgs_sddmm_csr) , dloop = gloop[i]
//   dsm_k  = dal_k * keep_k * scale ;  dot = sum_k sm_k dsm_k (incl. loop)
//   de_k   = sm_k (dsm_k - dot) ;  dpre_k = de_k * (pre_k > 0 ? 1 : slope)
//   ge[eid_k] = dpre_k (-> d a_s[src_k], summed per source by the caller) ; d a_d[i] = sum_k dpre_k ;
//   gsl[i] = dpre_loop  (the loop's contribution to d a_s[i])
__global__ void __launch_bounds__(kT) gat_alpha_bwd(const float* __restrict__ a_s, const float* __restrict__ a_d, int64_t N,
                                                   const int* __restrict__ in_ptr, const int* __restrict__ in_src,
                                                   const int* __restrict__ in_eid, float slope, float drop_scale,
                                                   uint32_t drop_thresh, int use_drop, uint64_t seed, uint32_t site,
                                                   const uint64_t* __restrict__ epoch,
                                                   const float* __restrict__ soft_in, const float* __restrict__ soft_loop,
                                                   const float* __restrict__ galpha, const float* __restrict__ gloop,
                                                   float* __restrict__ ge, float* __restrict__ gsl, float* __restrict__ d_ad) {
    seed = fold_epoch(seed, epoch);
    const int lane = threadIdx.x & 63;
    const int64_t i = (static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x) >> 6;
    if (i >= N) return;
    const int b = in_ptr[i], e = in_ptr[i + 1];
    const float ad = a_d[i];
    auto dsm_edge = [&](int k) {
        float g = galpha[in_eid[k]];
        if (use_drop) g = dropout_keep_at(seed, site, static_cast<uint64_t>(in_eid[k]), 0u, drop_thresh) ? g * drop_scale : 0.f;
        return g;
    };
    float gl = gloop[i];
    if (use_drop) gl = dropout_keep_at(seed, site + 1u, static_cast<uint64_t>(i), 0u, drop_thresh) ? gl * drop_scale : 0.f;
    float dot = 0.f;
    for (int k = b + lane; k < e; k += 64)
        if (in_src[k] != static_cast<int>(i)) dot += soft_in[k] * dsm_edge(k);
    dot = wave_sum_all(dot) + soft_loop[i] * gl;
    float dad = 0.f;
    for (int k = b + lane; k < e; k += 64) {
        const int s = in_src[k];
        float dpre = 0.f;
        if (s != static_cast<int>(i)) {
            const float pre = a_s[s] + ad;
            dpre = soft_in[k] * (dsm_edge(k) - dot) * (pre > 0.f ? 1.f : slope);
        }
        ge[in_eid[k]] = dpre;
        dad += dpre;
    }
    dad = wave_sum_all(dad);
    if (lane == 0) {
        const float pre = a_s[i] + ad;
        const float dl = soft_loop[i] * (gl - dot) * (pre > 0.f ? 1.f : slope);
        gsl[i] = dl;
        d_ad[i] = dad + dl;
    }
}

// out_order[k] = by_eid[eid[k]]  (re-order a per-edge array into a CSR's entry order)
// `dyn` (sgs_dyn_edges_set): the entries of a staged parent CSR past the live edge count are stale -- n = min(n, *dyn).  (A drawn
// subgraph has q < live edges, so its re-orderings are untouched.)
__global__ void __launch_bounds__(kT) gather_by_eid(const float* __restrict__ by_eid, const int* __restrict__ eid, int64_t n,
                                                   float* __restrict__ out_order, const int64_t* __restrict__ dyn) {
    const int64_t k = static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x;
    if (dyn && *dyn < n) n = *dyn;
    if (k < n) out_order[k] = by_eid[eid[k]];
}
__global__ void __launch_bounds__(kT) scatter_by_eid(const float* __restrict__ in_order, const int* __restrict__ eid, int64_t n,
                                                    float* __restrict__ by_eid, const int64_t* __restrict__ dyn) {
    const int64_t k = static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x;
    if (dyn && *dyn < n) n = *dyn;
    if (k < n) by_eid[eid[k]] = in_order[k];
}

// Node-level attention scores a_s = <x', att_src>, a_d = <x', att_dst> in ONE pass over x' (one wave per row, 16-byte loads).  The library
// GEMV the host used for these ran at 280 us per call on [33 869 x 256] (rocprofv3, config 4: 4 calls per step, a fifth of the step).
__global__ void __launch_bounds__(kT) gat_scores_fwd(const float* __restrict__ xl, int64_t N, int64_t D, const float* __restrict__ att_s,
                                                    const float* __restrict__ att_d, float* __restrict__ a_s, float* __restrict__ a_d) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x) >> 6;
    if (i >= N) return;
    float s = 0.f, d = 0.f;
    if ((D & 3) == 0) {
        for (int64_t c = 4 * lane; c < D; c += 256) {
            const float4 x = *reinterpret_cast<const float4*>(xl + i * D + c);
            const float4 u = *reinterpret_cast<const float4*>(att_s + c);
            const float4 v = *reinterpret_cast<const float4*>(att_d + c);
            s = fmaf(x.x, u.x, fmaf(x.y, u.y, fmaf(x.z, u.z, fmaf(x.w, u.w, s))));
            d = fmaf(x.x, v.x, fmaf(x.y, v.y, fmaf(x.z, v.z, fmaf(x.w, v.w, d))));
        }
    } else {
        for (int64_t c = lane; c < D; c += 64) {
            const float x = xl[i * D + c];
            s = fmaf(x, att_s[c], s);
            d = fmaf(x, att_d[c], d);
        }
    }
    s = wave_sum_all(s);
    d = wave_sum_all(d);
    if (lane == 0) { a_s[i] = s; a_d[i] = d; }
}

// backward: dxl[i, :] (+)= g_s[i] att_s + g_d[i] att_d  (ACC: added to an existing gradient);  per-workgroup partial sums of
// d att_s = sum_i g_s[i] x'[i, :], d att_d likewise (part [gridDim.x][2][D]; a second tiny launch adds them in order)
template <bool ACC>
__global__ void __launch_bounds__(kT) gat_scores_bwd(const float* __restrict__ xl, int64_t N, int64_t D, const float* __restrict__ att_s,
                                                    const float* __restrict__ att_d, const float* __restrict__ g_s, const float* __restrict__ g_d,
                                                    float* __restrict__ dxl, float* __restrict__ part, int rows_per_wg) {
    // thread t owns column t (D <= 256 per pass); a workgroup walks rows_per_wg rows
    const int64_t r0 = static_cast<int64_t>(blockIdx.x) * rows_per_wg;
    const int64_t r1 = r0 + rows_per_wg < N ? r0 + rows_per_wg : N;
    for (int64_t cb = 0; cb < D; cb += kT) {
        const int64_t c = cb + threadIdx.x;
        const bool in = c < D;
        const float us = in ? att_s[c] : 0.f, ud = in ? att_d[c] : 0.f;
        float ps = 0.f, pd = 0.f;
        for (int64_t i = r0; i < r1; ++i) {
            const float gs = g_s[i], gd = g_d[i];
            if (in) {
                const float x = xl[i * D + c];
                ps = fmaf(gs, x, ps);
                pd = fmaf(gd, x, pd);
                const float v = fmaf(gs, us, gd * ud);
                dxl[i * D + c] = ACC ? dxl[i * D + c] + v : v;
            }
        }
        if (in) {
            part[(static_cast<int64_t>(blockIdx.x) * 2) * D + c] = ps;
            part[(static_cast<int64_t>(blockIdx.x) * 2 + 1) * D + c] = pd;
        }
    }
}

// d att = sum over the workgroups' partial rows, fixed order: 64 columns x 16 row groups per workgroup (row group g takes partials g, g + 16,
// ... with four independent running sums; the groups are then added in order).  One thread per column walking all `nwg` rows was a chain of
// 530 dependent loads at arxiv-year's size: 89 us for 1 MB.
__global__ void __launch_bounds__(1024) gat_scores_bwd_finish(const float* __restrict__ part, int nwg, int64_t D, float* __restrict__ datt_s,
                                                             float* __restrict__ datt_d) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int64_t c = static_cast<int64_t>(blockIdx.x) * 64 + lane;             // column of the concatenated [2 D] vector
    const bool ok = c < 2 * D;
    const int which = c >= D ? 1 : 0;
    const int64_t cc = which ? c - D : c;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (ok) {
        const float* col = part + static_cast<int64_t>(which) * D + cc;
        int w = g;
        for (; w + 48 < nwg; w += 64) {
            a0 += col[static_cast<int64_t>(w) * 2 * D];
            a1 += col[static_cast<int64_t>(w + 16) * 2 * D];
            a2 += col[static_cast<int64_t>(w + 32) * 2 * D];
            a3 += col[static_cast<int64_t>(w + 48) * 2 * D];
        }
        for (; w < nwg; w += 16) a0 += col[static_cast<int64_t>(w) * 2 * D];
    }
    red[g][lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (g == 0 && ok) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) acc += red[k][lane];
        (which ? datt_d : datt_s)[cc] = acc;
    }
}

}  // namespace
}  // namespace sgs

using namespace sgs;

extern "C" {

int sgs_gat_alpha_fwd(const float* a_src, const float* a_dst, int64_t N, int64_t n_edges, const int32_t* in_ptr,
                      const int32_t* in_src, const int32_t* in_eid, float negative_slope, float p_drop, uint64_t seed,
                      uint32_t site, float* soft_in, float* soft_loop, float* alpha_in, float* alpha_loop,
                      sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(N >= 0 && n_edges >= 0 && p_drop >= 0.f && p_drop < 1.f, SGS_EINVAL, "sgs_gat_alpha_fwd: bad arguments");
    if (N == 0) return SGS_OK;
    SGS_REQUIRE(a_src && a_dst && in_ptr && soft_loop && alpha_loop && (n_edges == 0 || (in_src && in_eid && soft_in && alpha_in)),
                SGS_EINVAL, "sgs_gat_alpha_fwd: null pointer");
    hipLaunchKernelGGL(gat_alpha_fwd, dim3(cdiv(N * 64, kT)), dim3(kT), 0, stream, a_src, a_dst, N, in_ptr, in_src, in_eid,
                       negative_slope, 1.0f / (1.0f - p_drop), dropout_thresh(p_drop), p_drop > 0.f ? 1 : 0, seed, site, epoch_ptr(), soft_in,
                       soft_loop, alpha_in, alpha_loop);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

int sgs_gat_alpha_bwd(const float* a_src, const float* a_dst, int64_t N, int64_t n_edges, const int32_t* in_ptr,
                      const int32_t* in_src, const int32_t* in_eid, float negative_slope, float p_drop, uint64_t seed,
                      uint32_t site, const float* soft_in, const float* soft_loop, const float* galpha, const float* gloop,
                      float* g_edge, float* g_selfloop, float* d_a_dst, sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(N >= 0 && n_edges >= 0 && p_drop >= 0.f && p_drop < 1.f, SGS_EINVAL, "sgs_gat_alpha_bwd: bad arguments");
    if (N == 0) return SGS_OK;
    SGS_REQUIRE(a_src && a_dst && in_ptr && soft_loop && gloop && g_selfloop && d_a_dst, SGS_EINVAL, "sgs_gat_alpha_bwd: null pointer");
    hipLaunchKernelGGL(gat_alpha_bwd, dim3(cdiv(N * 64, kT)), dim3(kT), 0, stream, a_src, a_dst, N, in_ptr, in_src, in_eid,
                       negative_slope, 1.0f / (1.0f - p_drop), dropout_thresh(p_drop), p_drop > 0.f ? 1 : 0, seed, site, epoch_ptr(), soft_in,
                       soft_loop, galpha, gloop, g_edge, g_selfloop, d_a_dst);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

int sgs_gather_by_eid(const float* by_eid, const int32_t* eid, int64_t n, float* out_order, sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(n >= 0, SGS_EINVAL, "sgs_gather_by_eid: bad size");
    if (n == 0) return SGS_OK;
    SGS_REQUIRE(by_eid && eid && out_order, SGS_EINVAL, "sgs_gather_by_eid: null pointer");
    hipLaunchKernelGGL(gather_by_eid, dim3(cdiv(n, kT)), dim3(kT), 0, stream, by_eid, eid, n, out_order, dyn_edges_ptr());
    SGS_LAUNCH_OK();
    return SGS_OK;
}

int sgs_scatter_by_eid(const float* in_order, const int32_t* eid, int64_t n, float* by_eid, sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(n >= 0, SGS_EINVAL, "sgs_scatter_by_eid: bad size");
    if (n == 0) return SGS_OK;
    SGS_REQUIRE(in_order && eid && by_eid, SGS_EINVAL, "sgs_scatter_by_eid: null pointer");
    hipLaunchKernelGGL(scatter_by_eid, dim3(cdiv(n, kT)), dim3(kT), 0, stream, in_order, eid, n, by_eid, dyn_edges_ptr());
    SGS_LAUNCH_OK();
    return SGS_OK;
}

int sgs_gat_scores_fwd(const float* xl, int64_t N, int64_t D, const float* att_src, const float* att_dst, float* a_src, float* a_dst,
                       sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(N >= 0 && D > 0, SGS_EINVAL, "sgs_gat_scores_fwd: bad sizes");
    if (N == 0) return SGS_OK;
    SGS_REQUIRE(xl && att_src && att_dst && a_src && a_dst, SGS_EINVAL, "sgs_gat_scores_fwd: null pointer");
    hipLaunchKernelGGL(gat_scores_fwd, dim3(static_cast<unsigned>((N * 64 + kT - 1) / kT)), dim3(kT), 0, stream, xl, N, D, att_src, att_dst, a_src, a_dst);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

static inline int gat_scores_rows_per_wg(int64_t N) { return N >= 65536 ? 256 : (N >= 4096 ? 64 : 16); }
size_t sgs_gat_scores_bwd_workspace_bytes(int64_t N, int64_t D) {
    if (N < 0) N = 0;
    if (D < 0) D = 0;
    const int64_t rp = gat_scores_rows_per_wg(N);
    return static_cast<size_t>((N + rp - 1) / rp) * 2 * D * 4 + 256;
}

int sgs_gat_scores_bwd(const float* xl, int64_t N, int64_t D, const float* att_src, const float* att_dst, const float* g_src, const float* g_dst,
                       int accumulate, float* dxl, float* datt_src, float* datt_dst, void* ws, size_t ws_bytes, sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(N >= 0 && D > 0, SGS_EINVAL, "sgs_gat_scores_bwd: bad sizes");
    if (N == 0) return SGS_OK;
    SGS_REQUIRE(xl && att_src && att_dst && g_src && g_dst && dxl && datt_src && datt_dst, SGS_EINVAL, "sgs_gat_scores_bwd: null pointer");
    SGS_REQUIRE(ws && ws_bytes >= sgs_gat_scores_bwd_workspace_bytes(N, D), SGS_EWORKSPACE, "sgs_gat_scores_bwd: workspace too small");
    const int rp = gat_scores_rows_per_wg(N);
    const int nwg = static_cast<int>((N + rp - 1) / rp);
    float* part = static_cast<float*>(ws);
    if (accumulate)
        hipLaunchKernelGGL((gat_scores_bwd<true>), dim3(nwg), dim3(kT), 0, stream, xl, N, D, att_src, att_dst, g_src, g_dst, dxl, part, rp);
    else
        hipLaunchKernelGGL((gat_scores_bwd<false>), dim3(nwg), dim3(kT), 0, stream, xl, N, D, att_src, att_dst, g_src, g_dst, dxl, part, rp);
    hipLaunchKernelGGL(gat_scores_bwd_finish, dim3(static_cast<unsigned>((2 * D + 63) / 64)), dim3(1024), 0, stream, part, nwg, D, datt_src, datt_dst);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

}  // extern "C"
