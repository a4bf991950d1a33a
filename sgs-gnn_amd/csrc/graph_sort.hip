// CSR of both orientations of a LARGE edge list (>= kSortEdges entries) by two stable LSD radix sorts of (key = endpoint,
// value = edge id) -- sgs_graph_build's path for whole graphs (config 5: 114.6 M edges).  The counting / atomic-cursor / per-row
// sort path that serves METIS partitions is quadratic in the longest row (a hub with 3e5 in-edges: 1.5 s per build, measured,
// profiles/r02_s5_kernel_stats_before.csv) and its 2 x E global atomics cost 46 ms; a stable sort by endpoint yields exactly
// the required order -- rows by endpoint, entries of a row by ascending edge id -- with no atomics and no per-row pass:
// ceil(log2 N / 8) onesweep passes per orientation, each one streaming read + write of 8 B per edge.
// The device-wide radix sort itself is the ROCm library primitive (hipCUB / rocPRIM), as the vendor GEMM is for X W^T.
#include <hipcub/hipcub.hpp>

#include "sgs_common.h"

namespace sgs {
namespace {
constexpr int kT = 256;

// keys of both orientations + edge ids; existing self loops -> loop_eid[i] = LAST (i, i) edge id (PyG: last one wins)
__global__ void __launch_bounds__(kT) extract_endpoints(const int64_t* __restrict__ ei, int64_t n, int* __restrict__ ksrc, int* __restrict__ kdst,
                                                       int* __restrict__ vals, int* __restrict__ loop_eid) {
    const int64_t e = static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x;
    if (e >= n) return;
    const int s = static_cast<int>(ei[e]), d = static_cast<int>(ei[n + e]);
    ksrc[e] = s;
    kdst[e] = d;
    vals[e] = static_cast<int>(e);
    if (s == d) atomicMax(&loop_eid[s], static_cast<int>(e));          // integer max: order-independent
}

// ptr[i] = first position whose key is >= i (i = 0 .. N), keys ascending
__global__ void __launch_bounds__(kT) ptr_from_sorted(const int* __restrict__ keys, int64_t n, int64_t N, int* __restrict__ ptr) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x;
    if (i > N) return;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < i) lo = mid + 1; else hi = mid;
    }
    ptr[i] = static_cast<int>(lo);
}

// other[k] = the OTHER endpoint of edge eid[k] (row `other_row` of edge_index)
__global__ void __launch_bounds__(kT) gather_other(const int64_t* __restrict__ ei_row, const int* __restrict__ eid, int64_t n, int* __restrict__ other) {
    const int64_t k = static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x;
    if (k < n) other[k] = static_cast<int>(ei_row[eid[k]]);
}

int key_bits(int64_t N) {
    int b = 1;
    while ((int64_t(1) << b) < N) ++b;
    return b;
}

size_t cub_temp_bytes(int64_t n, int64_t N) {
    size_t bytes = 0;
    const int* kin = nullptr;
    int* kout = nullptr;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, kin, kout, kin, kout, static_cast<int>(n), 0, key_bits(N), nullptr);
    return bytes;
}
}  // namespace

size_t graph_sort_workspace_bytes(int64_t n, int64_t N) {
    return 4 * carve_bytes(static_cast<size_t>(n) + 1, 4) + carve_bytes(cub_temp_bytes(n, N) + 256, 1) + 512;
}

int graph_build_by_sort(const int64_t* ei, int64_t n, int64_t N, int32_t* in_ptr, int32_t* in_src, int32_t* in_eid, int32_t* out_ptr,
                        int32_t* out_dst, int32_t* out_eid, int32_t* loop_eid, void* ws, size_t ws_bytes, hipStream_t stream) {
    SGS_REQUIRE(ws_bytes >= graph_sort_workspace_bytes(n, N), SGS_EWORKSPACE, "sgs_graph_build: workspace too small for the sort path");
    Carver cv(ws);
    int* ksrc = cv.take<int>(n + 1);
    int* kdst = cv.take<int>(n + 1);
    int* vals = cv.take<int>(n + 1);
    int* ksorted = cv.take<int>(n + 1);
    size_t temp_bytes = cub_temp_bytes(n, N);
    void* temp = cv.take<char>(temp_bytes + 256);
    const int bits = key_bits(N);
    if (int rc = fill2_async(loop_eid, static_cast<size_t>(N) * 4, 0xFFFFFFFFu, loop_eid, 0, 0u, stream)) return rc;
    const dim3 ge(static_cast<unsigned>(cdiv(n, kT))), gn(static_cast<unsigned>(cdiv(N + 1, kT))), blk(kT);
    hipLaunchKernelGGL(extract_endpoints, ge, blk, 0, stream, ei, n, ksrc, kdst, vals, loop_eid);
    // in-CSR: rows = dst
    SGS_HIP_OK(hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, static_cast<const int*>(kdst), ksorted, static_cast<const int*>(vals), in_eid,
                                                  static_cast<int>(n), 0, bits, stream));
    hipLaunchKernelGGL(ptr_from_sorted, gn, blk, 0, stream, ksorted, n, N, in_ptr);
    hipLaunchKernelGGL(gather_other, ge, blk, 0, stream, ei, in_eid, n, in_src);
    // out-CSR: rows = src
    SGS_HIP_OK(hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, static_cast<const int*>(ksrc), ksorted, static_cast<const int*>(vals), out_eid,
                                                  static_cast<int>(n), 0, bits, stream));
    hipLaunchKernelGGL(ptr_from_sorted, gn, blk, 0, stream, ksorted, n, N, out_ptr);
    hipLaunchKernelGGL(gather_other, ge, blk, 0, stream, ei + n, out_eid, n, out_dst);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

}  // namespace sgs
