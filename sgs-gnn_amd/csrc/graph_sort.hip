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

// ---- edge lists already SORTED BY SOURCE (a draw over a row-sorted edge list keeps its order): the out-CSR is the list itself, only the
// in-CSR needs a sort -- of (key = dst, value = (src, edge id) packed in 64 bits), so that no gather follows it.
__global__ void __launch_bounds__(kT) extract_src_sorted(const int64_t* __restrict__ ei, int64_t n, int* __restrict__ kdst, uint64_t* __restrict__ vals,
                                                        int* __restrict__ out_dst, int* __restrict__ out_eid, int* __restrict__ loop_eid,
                                                        int* __restrict__ unsorted) {
    const int64_t e = static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x;
    if (e >= n) return;
    const int64_t s64 = ei[e];
    const int s = static_cast<int>(s64), d = static_cast<int>(ei[n + e]);
    if (unsorted && e + 1 < n && ei[e + 1] < s64) *unsorted = 1;                    // the precondition, checked where the data passes anyway
    kdst[e] = d;
    vals[e] = (static_cast<uint64_t>(static_cast<uint32_t>(s)) << 32) | static_cast<uint32_t>(e);
    out_dst[e] = d;
    out_eid[e] = static_cast<int>(e);
    if (s == d) atomicMax(&loop_eid[s], static_cast<int>(e));
}
// ptr[i] = first position whose source is >= i, on the int64 source row itself
__global__ void __launch_bounds__(kT) ptr_from_sorted64(const int64_t* __restrict__ keys, int64_t n, int64_t N, int* __restrict__ ptr) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x;
    if (i > N) return;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < i) lo = mid + 1; else hi = mid;
    }
    ptr[i] = static_cast<int>(lo);
}
__global__ void __launch_bounds__(kT) unpack_src_eid(const uint64_t* __restrict__ vals, int64_t n, int* __restrict__ in_src, int* __restrict__ in_eid) {
    const int64_t k = static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x;
    if (k >= n) return;
    const uint64_t v = vals[k];
    in_src[k] = static_cast<int>(v >> 32);
    in_eid[k] = static_cast<int>(v & 0xFFFFFFFFu);
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
size_t cub_temp_bytes64(int64_t n, int64_t N) {
    size_t bytes = 0;
    const int* kin = nullptr;
    int* kout = nullptr;
    const uint64_t* vin = nullptr;
    uint64_t* vout = nullptr;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, kin, kout, vin, vout, static_cast<int>(n), 0, key_bits(N), nullptr);
    return bytes;
}
}  // namespace

size_t graph_src_sorted_workspace_bytes(int64_t n, int64_t N) {
    return 2 * carve_bytes(static_cast<size_t>(n) + 1, 4) + 2 * carve_bytes(static_cast<size_t>(n) + 1, 8) + carve_bytes(cub_temp_bytes64(n, N) + 256, 1) + 768;
}

// `unsorted_flag` (device word): set to 1 when the list turns out NOT to be sorted by source -- the arrays are then meaningless
int graph_build_src_sorted(const int64_t* ei, int64_t n, int64_t N, int32_t* in_ptr, int32_t* in_src, int32_t* in_eid, int32_t* out_ptr,
                           int32_t* out_dst, int32_t* out_eid, int32_t* loop_eid, int32_t* unsorted_flag, void* ws, size_t ws_bytes,
                           hipStream_t stream) {
    SGS_REQUIRE(ws_bytes >= graph_src_sorted_workspace_bytes(n, N), SGS_EWORKSPACE, "sgs_graph_build_src_sorted: workspace too small");
    Carver cv(ws);
    int* kdst = cv.take<int>(n + 1);
    int* ksorted = cv.take<int>(n + 1);
    uint64_t* vals = cv.take<uint64_t>(n + 1);
    uint64_t* vsorted = cv.take<uint64_t>(n + 1);
    size_t temp_bytes = cub_temp_bytes64(n, N);
    void* temp = cv.take<char>(temp_bytes + 256);
    const int bits = key_bits(N);
    if (int rc = fill2_async(loop_eid, static_cast<size_t>(N) * 4, 0xFFFFFFFFu, unsorted_flag ? unsorted_flag : loop_eid, unsorted_flag ? 4 : 0, 0u, stream)) return rc;
    const dim3 ge(static_cast<unsigned>(cdiv(n, kT))), gn(static_cast<unsigned>(cdiv(N + 1, kT))), blk(kT);
    if (n > 0) hipLaunchKernelGGL(extract_src_sorted, ge, blk, 0, stream, ei, n, kdst, vals, out_dst, out_eid, loop_eid, unsorted_flag);
    hipLaunchKernelGGL(ptr_from_sorted64, gn, blk, 0, stream, ei, n, N, out_ptr);
    if (n > 0)
        SGS_HIP_OK(hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, static_cast<const int*>(kdst), ksorted, static_cast<const uint64_t*>(vals), vsorted,
                                                      static_cast<int>(n), 0, bits, stream));
    hipLaunchKernelGGL(ptr_from_sorted, gn, blk, 0, stream, ksorted, n, N, in_ptr);
    if (n > 0) hipLaunchKernelGGL(unpack_src_eid, ge, blk, 0, stream, vsorted, n, in_src, in_eid);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

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
