// K4 / K5: graph preparation, gcn_norm and weighted CSR SpMM / SDDMM for the GCN layers (gfx950).
//
// Reference: PyG 2.3.1 GCNConv as called at model.py:94-95,107-111 (scorer encoder) and
// model.py:151-153,159-161 (GNNModel): add_remaining_self_loops -> deg = scatter_add(w, dst)
// -> w_hat = deg^-1/2[src] w deg^-1/2[dst] -> out[dst] += w_hat x'[src] -> + bias.
// The reference materialises [nnz, D] messages and scatter-adds them (atomics on GPU); here
// the aggregation is a gather over a dst-sorted CSR (no float atomics, deterministic), and
// the transposed pass of backward is the same kernel over the src-sorted CSR.
//
// All kernels are HBM/L2-bandwidth bound gathers: per nnz 4 B col + 4 B weight + 4*D B row.
#include "sgs_common.h"

namespace sgs {
size_t graph_sort_workspace_bytes(int64_t n, int64_t N);
size_t graph_src_sorted_workspace_bytes(int64_t n, int64_t N);
int graph_build_src_sorted(const int64_t* ei, int64_t n, int64_t N, int32_t* in_ptr, int32_t* in_src, int32_t* in_eid, int32_t* out_ptr,
                           int32_t* out_dst, int32_t* out_eid, int32_t* loop_eid, int32_t* unsorted_flag, void* ws, size_t ws_bytes,
                           hipStream_t stream);
int graph_build_by_sort(const int64_t* ei, int64_t n, int64_t N, int32_t* in_ptr, int32_t* in_src, int32_t* in_eid, int32_t* out_ptr,
                        int32_t* out_dst, int32_t* out_eid, int32_t* loop_eid, void* ws, size_t ws_bytes, hipStream_t stream);
}

namespace sgs {
namespace {

constexpr int kT = 256;
constexpr int kMaxLdsRow = 8192;   // longest row sorted in LDS (32 KiB)

// ---------------------------------------------------------------- CSR build
__global__ void __launch_bounds__(kT) count_degrees(const int64_t* __restrict__ ei, int64_t n_edges, int* __restrict__ cnt_in,
                                                   int* __restrict__ cnt_out, int* __restrict__ loop_eid) {
    const int64_t e = static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x;
    if (e >= n_edges) return;
    const int s = static_cast<int>(ei[e]), d = static_cast<int>(ei[n_edges + e]);
    atomicAdd(&cnt_in[d], 1);
    atomicAdd(&cnt_out[s], 1);
    if (s == d) atomicMax(&loop_eid[s], static_cast<int>(e));   // PyG: the last existing loop wins
}

// blockIdx.x = 0: in-direction, 1: out-direction.  Exclusive scan of N counts -> ptr[N+1]; also initialises the fill cursors.
// Tiles of 4096 consecutive counts, four per thread (coalesced), wave scan + the 16 wave totals through LDS, a running carry; the next
// tile's counts are loaded before this tile's barrier.  (Each thread scanning its own N / 1024 consecutive counts -- every access of a
// wave a different cache line -- took 70 us at N = 33 869.)
__global__ void __launch_bounds__(1024) scan_counts(const int* __restrict__ cnt_in, const int* __restrict__ cnt_out,
                                                   int64_t N, int* __restrict__ in_ptr, int* __restrict__ out_ptr,
                                                   int* __restrict__ cur_in, int* __restrict__ cur_out) {
    __shared__ int wtot[2][16];
    const int* cnt = blockIdx.x == 0 ? cnt_in : cnt_out;
    int* ptr = blockIdx.x == 0 ? in_ptr : out_ptr;
    int* cur = blockIdx.x == 0 ? cur_in : cur_out;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    auto load4 = [&](int64_t i, int (&v)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = i + j < N ? cnt[i + j] : 0;
    };
    int carry = 0, buf = 0;
    int v[4], vn[4];
    load4(static_cast<int64_t>(threadIdx.x) * 4, v);
    for (int64_t base = 0; base < N; base += 4096, buf ^= 1) {
        const int64_t i = base + static_cast<int64_t>(threadIdx.x) * 4;
        load4(i + 4096, vn);
        const int mine = (v[0] + v[1]) + (v[2] + v[3]);
        int inc = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(inc, o, 64);
            if (lane >= o) inc += t;
        }
        if (lane == 63) wtot[buf][wid] = inc;
        __syncthreads();                            // (two LDS rows alternate: one barrier per tile)
        int run = carry + inc - mine, total = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const int t = wtot[buf][w];
            run += w < wid ? t : 0;
            total += t;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (i + j < N) { ptr[i + j] = run; cur[i + j] = run; }
            run += v[j];
            v[j] = vn[j];
        }
        carry += total;
    }
    if (threadIdx.x == 0) ptr[N] = carry;
}

__global__ void __launch_bounds__(kT) fill_rows(const int64_t* __restrict__ ei, int64_t n_edges, int* __restrict__ cur_in,
                                               int* __restrict__ cur_out, int* __restrict__ tmp_in, int* __restrict__ tmp_out) {
    const int64_t e = static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x;
    if (e >= n_edges) return;
    const int s = static_cast<int>(ei[e]), d = static_cast<int>(ei[n_edges + e]);
    tmp_in[atomicAdd(&cur_in[d], 1)] = static_cast<int>(e);
    tmp_out[atomicAdd(&cur_out[s], 1)] = static_cast<int>(e);
}

// Small-N variants (N <= kLdsNodes, e.g. METIS partitions): every edge hammering ~1k global counters
// is contention-bound, so a workgroup first counts / ranks its chunk of edges in LDS and touches each
// global counter at most once per chunk.
constexpr int kLdsNodes = 4096;
constexpr int kEdgesPerBlock = 4096;      // 16 per thread
__global__ void __launch_bounds__(kT) count_degrees_lds(const int64_t* __restrict__ ei, int64_t n_edges, int N, int* __restrict__ cnt_in,
                                                       int* __restrict__ cnt_out, int* __restrict__ loop_eid) {
    __shared__ int lin[kLdsNodes], lout[kLdsNodes];
    for (int i = threadIdx.x; i < N; i += kT) { lin[i] = 0; lout[i] = 0; }
    __syncthreads();
    const int64_t base = static_cast<int64_t>(blockIdx.x) * kEdgesPerBlock;
    for (int it = 0; it < kEdgesPerBlock / kT; ++it) {
        const int64_t e = base + it * kT + threadIdx.x;
        if (e < n_edges) {
            const int s = static_cast<int>(ei[e]), d = static_cast<int>(ei[n_edges + e]);
            atomicAdd(&lin[d], 1);
            atomicAdd(&lout[s], 1);
            if (s == d) atomicMax(&loop_eid[s], static_cast<int>(e));
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < N; i += kT) {
        if (lin[i]) atomicAdd(&cnt_in[i], lin[i]);
        if (lout[i]) atomicAdd(&cnt_out[i], lout[i]);
    }
}

__global__ void __launch_bounds__(kT) fill_rows_lds(const int64_t* __restrict__ ei, int64_t n_edges, int N, int* __restrict__ cur_in,
                                                   int* __restrict__ cur_out, int* __restrict__ tmp_in, int* __restrict__ tmp_out) {
    __shared__ int lin[kLdsNodes], lout[kLdsNodes];
    for (int i = threadIdx.x; i < N; i += kT) { lin[i] = 0; lout[i] = 0; }
    __syncthreads();
    const int64_t base = static_cast<int64_t>(blockIdx.x) * kEdgesPerBlock;
    int rs[kEdgesPerBlock / kT], rd[kEdgesPerBlock / kT], ss[kEdgesPerBlock / kT], dd[kEdgesPerBlock / kT];
#pragma unroll
    for (int it = 0; it < kEdgesPerBlock / kT; ++it) {
        const int64_t e = base + it * kT + threadIdx.x;
        ss[it] = -1;
        if (e < n_edges) {
            ss[it] = static_cast<int>(ei[e]);
            dd[it] = static_cast<int>(ei[n_edges + e]);
            rd[it] = atomicAdd(&lin[dd[it]], 1);      // rank inside this chunk
            rs[it] = atomicAdd(&lout[ss[it]], 1);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < N; i += kT) {       // reserve this chunk's slots: one global atomic per touched node
        const int ci = lin[i], co = lout[i];
        lin[i] = ci ? atomicAdd(&cur_in[i], ci) : 0;
        lout[i] = co ? atomicAdd(&cur_out[i], co) : 0;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < kEdgesPerBlock / kT; ++it) {
        if (ss[it] >= 0) {
            const int e = static_cast<int>(base + it * kT + threadIdx.x);
            tmp_in[lin[dd[it]] + rd[it]] = e;
            tmp_out[lout[ss[it]] + rs[it]] = e;
        }
    }
}

// The atomic fill leaves each row's edge ids in arrival order; sorting them (unique ints) makes
// the CSR -- and every floating-point sum over a row -- deterministic.
// Rows of <= 64 entries: one wave, bitonic network through lane shuffles.
__device__ __forceinline__ void sort_rows_wave_body(int64_t block, const int* __restrict__ in_ptr, const int* __restrict__ out_ptr, int64_t N,
                                                    const int* __restrict__ tmp_in, const int* __restrict__ tmp_out,
                                                    const int64_t* __restrict__ ei, int64_t n_edges, int* __restrict__ in_eid,
                                                    int* __restrict__ in_src, int* __restrict__ out_eid,
                                                    int* __restrict__ out_dst) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (block * kT + threadIdx.x) >> 6;
    if (r >= 2 * N) return;
    const bool out = r >= N;
    const int64_t row = out ? r - N : r;
    const int* ptr = out ? out_ptr : in_ptr;
    const int base = ptr[row], d = ptr[row + 1] - base;
    if (d == 0 || d > 64) return;
    const int* tmp = out ? tmp_out : tmp_in;
    int v = lane < d ? tmp[base + lane] : 0x7fffffff;
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int o = __shfl_xor(v, j, 64);
            const bool keep_min = ((lane & j) == 0) == ((lane & k) == 0);
            v = keep_min ? min(v, o) : max(v, o);
        }
    }
    if (lane < d) {
        if (out) { out_eid[base + lane] = v; out_dst[base + lane] = static_cast<int>(ei[n_edges + v]); }
        else     { in_eid[base + lane] = v;  in_src[base + lane] = static_cast<int>(ei[v]); }
    }
}

// Rows of > 64 entries: one block per row; bitonic sort in LDS up to kMaxLdsRow entries,
// rank-by-counting straight from L2 beyond that (hub rows).
__device__ __forceinline__ void sort_rows_block_body(int64_t r, int* a, const int* __restrict__ in_ptr, const int* __restrict__ out_ptr, int64_t N,
                                                     const int* __restrict__ tmp_in, const int* __restrict__ tmp_out,
                                                     const int64_t* __restrict__ ei, int64_t n_edges, int* __restrict__ in_eid,
                                                     int* __restrict__ in_src, int* __restrict__ out_eid,
                                                     int* __restrict__ out_dst) {
    const bool out = r >= N;
    const int64_t row = out ? r - N : r;
    const int* ptr = out ? out_ptr : in_ptr;
    const int base = ptr[row], d = ptr[row + 1] - base;
    if (d <= 64) return;
    const int* tmp = (out ? tmp_out : tmp_in) + base;
    int* eid_o = (out ? out_eid : in_eid) + base;
    int* col_o = (out ? out_dst : in_src) + base;
    const int64_t* colsrc = out ? ei + n_edges : ei;
    if (d <= kMaxLdsRow) {
        int n2 = 128;
        while (n2 < d) n2 <<= 1;
        for (int i = threadIdx.x; i < n2; i += kT) a[i] = i < d ? tmp[i] : 0x7fffffff;
        __syncthreads();
        for (int k = 2; k <= n2; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int idx = threadIdx.x; idx < (n2 >> 1); idx += kT) {
                    const int i = ((idx & ~(j - 1)) << 1) | (idx & (j - 1));
                    const int p = i | j;
                    const int x = a[i], y = a[p];
                    const bool asc = (i & k) == 0;
                    if ((x > y) == asc) { a[i] = y; a[p] = x; }
                }
                __syncthreads();
            }
        }
        for (int i = threadIdx.x; i < d; i += kT) {
            const int v = a[i];
            eid_o[i] = v;
            col_o[i] = static_cast<int>(colsrc[v]);
        }
    } else {
        for (int i = threadIdx.x; i < d; i += kT) {
            const int v = tmp[i];
            int rank = 0;
            for (int j = 0; j < d; ++j) rank += (tmp[j] < v);
            eid_o[rank] = v;
            col_o[rank] = static_cast<int>(colsrc[v]);
        }
    }
}

// One launch for both row classes: the first `n_wave_blocks` workgroups sort the short rows (one wave each), the
// remaining 2N workgroups the long ones (one workgroup each; short rows return at once).
__global__ void __launch_bounds__(kT) sort_rows(int64_t n_wave_blocks, const int* __restrict__ in_ptr, const int* __restrict__ out_ptr, int64_t N,
                                               const int* __restrict__ tmp_in, const int* __restrict__ tmp_out,
                                               const int64_t* __restrict__ ei, int64_t n_edges, int* __restrict__ in_eid,
                                               int* __restrict__ in_src, int* __restrict__ out_eid, int* __restrict__ out_dst) {
    __shared__ int a[kMaxLdsRow];
    if (static_cast<int64_t>(blockIdx.x) < n_wave_blocks)
        sort_rows_wave_body(blockIdx.x, in_ptr, out_ptr, N, tmp_in, tmp_out, ei, n_edges, in_eid, in_src, out_eid, out_dst);
    else
        sort_rows_block_body(static_cast<int64_t>(blockIdx.x) - n_wave_blocks, a, in_ptr, out_ptr, N, tmp_in, tmp_out, ei, n_edges, in_eid,
                             in_src, out_eid, out_dst);
}

// ---------------------------------------------------------------- CSR of a SAMPLED subgraph from its parent's CSR
// A drawn subgraph keeps a subset of the parent's edges in their original order, so its CSR (both orientations, rows
// sorted by edge id) is the parent's CSR with the unselected entries squeezed out and the edge ids renumbered by rank:
// no atomics, no per-row sort.  Three launches instead of the five of sgs_graph_build, and none of them is the row sort
// (19 us at partition scale):  (1) one wave per parent row counts its selected entries; spare workgroups scatter
// pos[sampled_eid[j]] = j;  (2) scan_counts;  (3) one wave per parent row compacts its entries with a ballot prefix.
// Large parents (>= 8 M edges): the byte mask (E bytes) is first packed into a bit mask (E/8 bytes) so that the random
// `mask[eid]` lookups along the in-rows hit a 14 MB table (L2 / Infinity Cache resident) instead of a 114 MB one.
__global__ void __launch_bounds__(kT) pack_mask_bits(const uint8_t* __restrict__ mask, int64_t E, uint32_t* __restrict__ bits) {
    const int64_t w = static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x;        // one 32-edge word per thread
    if (w * 32 >= E) return;
    uint32_t v = 0;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const int64_t e = w * 32 + j;
        if (e < E && mask[e]) v |= 1u << j;
    }
    bits[w] = v;
}
template <bool BITS>
__device__ __forceinline__ bool mask_at(const uint8_t* __restrict__ mask, int e) {
    if (BITS) return (reinterpret_cast<const uint32_t*>(mask)[e >> 5] >> (e & 31)) & 1u;
    return mask[e] != 0;
}

template <bool BITS>
__global__ void __launch_bounds__(kT) filter_count_scatter(const int* __restrict__ in_ptr, const int* __restrict__ in_eid,
                                                          const int* __restrict__ out_ptr, const int* __restrict__ out_eid, int64_t N,
                                                          const uint8_t* __restrict__ mask, const int64_t* __restrict__ sampled_eid,
                                                          int64_t q, int64_t n_row_blocks, int* __restrict__ cnt_in,
                                                          int* __restrict__ cnt_out, int* __restrict__ pos) {
    if (static_cast<int64_t>(blockIdx.x) >= n_row_blocks) {
        const int64_t j = (static_cast<int64_t>(blockIdx.x) - n_row_blocks) * kT + threadIdx.x;
        if (j < q) pos[sampled_eid[j]] = static_cast<int>(j);
        return;
    }
    const int lane = threadIdx.x & 63;
    const int64_t r = (static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x) >> 6;
    if (r >= 2 * N) return;
    const bool out = r >= N;
    const int64_t row = out ? r - N : r;
    const int* ptr = out ? out_ptr : in_ptr;
    const int* eid = out ? out_eid : in_eid;
    int c = 0;
    const int b = ptr[row], e = ptr[row + 1];
    for (int k = b + lane; k < e; k += 256) {          // four entries per lane in flight: ids first, then their mask words
        int id[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) id[u] = eid[min(k + 64 * u, e - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool m = mask_at<BITS>(mask, id[u]);
            c += (k + 64 * u < e && m) ? 1 : 0;
        }
    }
    c = wave_sum_int_all(c);
    if (lane == 0) (out ? cnt_out : cnt_in)[row] = c;
}

// SCAN: the child row pointers are not there yet -- every wave sums the counts of the rows before its own (N <= kSelfScanRows: a few
// cached loads per lane) and writes its entry of the pointer array (the last row also the total): the scan launch between the count and
// the fill pass goes away (partition scale: one launch = ~5 us of a ~90 us draw-to-normalisation chain).
constexpr int64_t kSelfScanRows = 4096;
template <bool BITS, bool SCAN = false>
__global__ void __launch_bounds__(kT) filter_fill(const int* __restrict__ pin_ptr, const int* __restrict__ pin_src, const int* __restrict__ pin_eid,
                                                 const int* __restrict__ pout_ptr, const int* __restrict__ pout_dst,
                                                 const int* __restrict__ pout_eid, int64_t N, const uint8_t* __restrict__ mask,
                                                 const int* __restrict__ pos, int* __restrict__ in_ptr, int* __restrict__ in_src,
                                                 int* __restrict__ in_eid, int* __restrict__ out_ptr, int* __restrict__ out_dst,
                                                 int* __restrict__ out_eid, int* __restrict__ loop_eid, const int* __restrict__ cnt_in = nullptr,
                                                 const int* __restrict__ cnt_out = nullptr) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x) >> 6;
    if (r >= 2 * N) return;
    const bool out = r >= N;
    const int64_t row = out ? r - N : r;
    const int* pptr = out ? pout_ptr : pin_ptr;
    const int* pcol = out ? pout_dst : pin_src;
    const int* peid = out ? pout_eid : pin_eid;
    int* ccol = out ? out_dst : in_src;
    int* ceid = out ? out_eid : in_eid;
    int w;                                            // write cursor of the child row
    if constexpr (SCAN) {
        const int* cnt = out ? cnt_out : cnt_in;
        int acc = 0;
        for (int64_t i = lane; i < row; i += 64) acc += cnt[i];
        w = wave_sum_int_all(acc);
        if (lane == 0) {
            int* cptr = out ? out_ptr : in_ptr;
            cptr[row] = w;
            if (row == N - 1) cptr[N] = w + cnt[row];
        }
    } else {
        w = (out ? out_ptr : in_ptr)[row];
    }
    int loop = -1;
    const int b = pptr[row], e = pptr[row + 1];
    // the next 64 entries' (edge id, column) are loaded while this step's mask / position lookups are in flight, and the id / column / mask
    // loads are unconditional (clamped): as `if (k < e) { pe = ..; sel = mask[pe]; if (sel) { col = ..; ne = pos[pe]; } }` a step was three
    // dependent memory round trips
    int pe_c = 0, col_c = 0;
    if (e > b) { const int kc = min(b + lane, e - 1); pe_c = peid[kc]; col_c = pcol[kc]; }
    for (int k0 = b; k0 < e; k0 += 64) {
        const int k = k0 + lane;
        const int kn = min(k + 64, e - 1);
        const int pe_n = peid[kn], col_n = pcol[kn];
        const bool m_c = mask_at<BITS>(mask, pe_c);
        const bool sel = k < e && m_c;
        const int ne = sel ? pos[pe_c] : 0;             // selected entries only: on a whole graph `pos` is hundreds of MB of random 4-byte reads
        const int col = col_c;
        pe_c = pe_n; col_c = col_n;
        const unsigned long long bal = __ballot(sel);
        if (sel) {
            const int o = w + __popcll(bal & ((1ull << lane) - 1ull));
            ccol[o] = col;
            ceid[o] = ne;
        }
        if (!out) {                                     // PyG: the last existing self loop of a node carries its loop weight
            const unsigned long long lb = __ballot(sel && col == static_cast<int>(row));
            if (lb) loop = __shfl(ne, 63 - __clzll(lb), 64);
        }
        w += __popcll(bal);
    }
    if (!out && lane == 0) loop_eid[row] = loop;
}

// ---------------------------------------------------------------- gcn_norm forward
// One wave per node: deg_i = loopw_i + sum_{k in in-row i, src != i} w[eid_k]; dis = deg^-1/2.
__global__ void __launch_bounds__(kT) norm_deg(const float* __restrict__ w, int64_t N, const int* __restrict__ in_ptr,
                                              const int* __restrict__ in_src, const int* __restrict__ in_eid,
                                              const int* __restrict__ loop_eid, float* __restrict__ dis,
                                              float* __restrict__ loopw) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x) >> 6;
    if (i >= N) return;
    const int b = in_ptr[i], e = in_ptr[i + 1];
    float acc = 0.f;
    for (int k = b + lane; k < e; k += 64) {
        if (in_src[k] != static_cast<int>(i)) acc += w ? w[in_eid[k]] : 1.0f;
    }
    acc = wave_sum_all(acc);
    const int le = loop_eid[i];
    const float lw = (le >= 0 && w) ? w[le] : 1.0f;
    const float deg = lw + acc;
    float di = 1.0f / sqrtf(deg);           // deg.pow(-0.5)
    if (isinf(di)) di = 0.f;                // masked_fill(inf -> 0)
    if (lane == 0) { dis[i] = di; loopw[i] = lw; }
}

// Edge-sharded graphs: each rank sums the weights of ITS in-edges (degpart), the ranks all-reduce, and
// the normalisation continues from the global degree (the added self loop counts once: + 1).
__global__ void __launch_bounds__(kT) deg_partial(const float* __restrict__ w, int64_t N, const int* __restrict__ in_ptr,
                                                 const int* __restrict__ in_src, const int* __restrict__ in_eid,
                                                 float* __restrict__ degpart) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x) >> 6;
    if (i >= N) return;
    float acc = 0.f;
    for (int k = in_ptr[i] + lane; k < in_ptr[i + 1]; k += 64)
        if (in_src[k] != static_cast<int>(i)) acc += w ? w[in_eid[k]] : 1.0f;
    acc = wave_sum_all(acc);
    if (lane == 0) degpart[i] = acc;
}
__global__ void __launch_bounds__(kT) dis_from_degree(const float* __restrict__ degsum, int64_t N, float* __restrict__ dis,
                                                     float* __restrict__ loopw) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x;
    if (i >= N) return;
    float di = 1.0f / sqrtf(1.0f + degsum[i]);
    if (isinf(di)) di = 0.f;
    dis[i] = di;
    loopw[i] = 1.0f;
}
// Y = act(X + bias) with the same fused ReLU / counter-based dropout as the SpMM epilogue.
__global__ void __launch_bounds__(kT) bias_act(const float* __restrict__ X, const float* __restrict__ bias, int64_t N, int64_t D, int act,
                                              float drop_scale, uint32_t drop_thresh, uint64_t seed, uint32_t site,
                                              const uint64_t* __restrict__ epoch, float* __restrict__ Y, int64_t row_offset) {
    seed = fold_epoch(seed, epoch);
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x;
    if (idx >= N * D) return;
    const int64_t i = idx / D;
    const uint32_t c = static_cast<uint32_t>(idx - i * D);
    float y = X[idx];
    if (bias) y += bias[c];
    if (act != SGS_ACT_NONE) y = fmaxf(y, 0.f);
    if (act == SGS_ACT_RELU_DROPOUT) y = dropout_keep_at(seed, site, static_cast<uint64_t>(row_offset + i), c, drop_thresh) ? y * drop_scale : 0.f;
    Y[idx] = y;
}

// GraphSAGE mean aggregation (PyG SAGEConv aggr='mean': mean over in-edges, duplicates and (i,i) edges
// counted, no loops added): entry weight 1 / indeg(dst) in both CSR orders.
__global__ void __launch_bounds__(kT) mean_weights(int64_t N, const int* __restrict__ in_ptr, const int* __restrict__ out_ptr,
                                                  const int* __restrict__ out_dst, float* __restrict__ what_in,
                                                  float* __restrict__ what_out) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x) >> 6;
    if (r >= 2 * N) return;
    if (r < N) {
        const int b = in_ptr[r], e = in_ptr[r + 1];
        const float w = e > b ? 1.0f / static_cast<float>(e - b) : 0.f;
        for (int k = b + lane; k < e; k += 64) what_in[k] = w;
    } else {
        const int64_t i = r - N;
        for (int k = out_ptr[i] + lane; k < out_ptr[i + 1]; k += 64) {
            const int t = out_dst[k];
            what_out[k] = 1.0f / static_cast<float>(in_ptr[t + 1] - in_ptr[t]);
        }
    }
}

// datasets.py:141-156 `add_degree` (before its softmax): logit_e = E^-1/2 / (colcount[row_e] + rowcount[col_e] + 1e-10),
// colcount = in-degree, rowcount = out-degree, read off the two CSR pointer arrays.
__global__ void __launch_bounds__(kT) degree_prior_logits(const int64_t* __restrict__ ei, int64_t E, const int* __restrict__ in_ptr,
                                                         const int* __restrict__ out_ptr, float scale, float* __restrict__ logit) {
    const int64_t e = static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x;
    if (e >= E) return;
    const int r = static_cast<int>(ei[e]), c = static_cast<int>(ei[E + e]);
    const float colcount_r = static_cast<float>(in_ptr[r + 1] - in_ptr[r]);
    const float rowcount_c = static_cast<float>(out_ptr[c + 1] - out_ptr[c]);
    const float prob = 1.0f / ((colcount_r + rowcount_c) + 1e-10f);
    logit[e] = prob * scale;
}

// Normalised weights in both CSR orders (0 for loop entries, which the loop term replaces).
__global__ void __launch_bounds__(kT) norm_weights(const float* __restrict__ w, int64_t N, int64_t n_edges,
                                                  const int* __restrict__ in_ptr, const int* __restrict__ in_src,
                                                  const int* __restrict__ in_eid, const int* __restrict__ out_ptr,
                                                  const int* __restrict__ out_dst, const int* __restrict__ out_eid,
                                                  const float* __restrict__ dis, const float* __restrict__ loopw,
                                                  float* __restrict__ what_in, float* __restrict__ what_out,
                                                  float* __restrict__ what_loop) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x) >> 6;
    if (r >= 2 * N) return;
    const bool out = r >= N;
    const int i = static_cast<int>(out ? r - N : r);
    const float di = dis[i];
    if (!out) {
        const int b = in_ptr[i], e = in_ptr[i + 1];
        for (int k = b + lane; k < e; k += 64) {
            const int s = in_src[k];
            const float we = w ? w[in_eid[k]] : 1.0f;
            what_in[k] = (s == i) ? 0.f : (dis[s] * we) * di;     // dis[row] * w * dis[col]
        }
        if (lane == 0) what_loop[i] = (di * loopw[i]) * di;
    } else {
        const int b = out_ptr[i], e = out_ptr[i + 1];
        for (int k = b + lane; k < e; k += 64) {
            const int t = out_dst[k];
            const float we = w ? w[out_eid[k]] : 1.0f;
            what_out[k] = (t == i) ? 0.f : (di * we) * dis[t];
        }
    }
}

// ---------------------------------------------------------------- gcn_norm backward
__global__ void __launch_bounds__(kT) norm_bwd_node(const float* __restrict__ w, const float* __restrict__ gw,
                                                   const float* __restrict__ gloop, int64_t N, const int* __restrict__ in_ptr,
                                                   const int* __restrict__ in_src, const int* __restrict__ in_eid,
                                                   const int* __restrict__ out_ptr, const int* __restrict__ out_dst,
                                                   const int* __restrict__ out_eid, const float* __restrict__ dis,
                                                   const float* __restrict__ loopw, float* __restrict__ Hn,
                                                   const float* __restrict__ gw2 = nullptr, const float* __restrict__ gloop2 = nullptr) {
    // gw2 / gloop2 (optional): a second layer's gradient wrt the same normalised weights, summed on read (both GCN layers of a model
    // share one normalisation: autograd's add kernel between the two SDDMMs and this pass is not needed)
    const int lane = threadIdx.x & 63;
    const int64_t t = (static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x) >> 6;
    if (t >= N) return;
    float acc = 0.f;
    // (unconditional clamped loads, the condition applied to the finished product: index pair, then the three gathers, per step)
    auto side = [&](const int* __restrict__ ptr, const int* __restrict__ col, const int* __restrict__ eid) {
        const int b = ptr[t], e_ = ptr[t + 1];
        for (int k0 = b; k0 < e_; k0 += 64) {
            const int k = k0 + lane, kc = min(k, e_ - 1);
            const int s = col[kc], e = eid[kc];
            const float ge = gw2 ? gw[e] + gw2[e] : gw[e];
            const float v = ge * w[e] * dis[s];
            if (k < e_ && s != static_cast<int>(t)) acc += v;
        }
    };
    side(in_ptr, in_src, in_eid);
    side(out_ptr, out_dst, out_eid);
    acc = wave_sum_all(acc);
    if (lane == 0) {
        const float a = dis[t];
        const float gl = gloop2 ? gloop[t] + gloop2[t] : gloop[t];
        const float G = acc + 2.0f * gl * loopw[t] * a;
        Hn[t] = -0.5f * a * a * a * G;
    }
}

__global__ void __launch_bounds__(kT) norm_bwd_edge(const float* __restrict__ gw, const float* __restrict__ gloop,
                                                   const int64_t* __restrict__ ei, int64_t n_edges,
                                                   const int* __restrict__ loop_eid, const float* __restrict__ dis,
                                                   const float* __restrict__ Hn, float* __restrict__ dw,
                                                   const float* __restrict__ gw2 = nullptr, const float* __restrict__ gloop2 = nullptr,
                                                   const float* __restrict__ dw_add = nullptr) {
    // dw_add (optional, may alias dw): another consumer's gradient wrt the same edge weights, added on the way out
    const int64_t e = static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x;
    if (e >= n_edges) return;
    const int s = static_cast<int>(ei[e]), t = static_cast<int>(ei[n_edges + e]);
    float g;
    if (s != t) g = (gw2 ? gw[e] + gw2[e] : gw[e]) * dis[s] * dis[t] + Hn[t];
    else g = (gloop2 ? gloop[s] + gloop2[s] : gloop[s]) * dis[s] * dis[s] + Hn[s];   // every existing (i,i) edge: PyG's index_put backward hands
                                                                                     // the loop gradient to overwritten duplicates too
    dw[e] = dw_add ? g + dw_add[e] : g;
}

// ---------------------------------------------------------------- CSR SpMM:  Y[i,:] = act( sum_k val[k] X[col[k],:] + diag[i] X[i,:] + bias )
// A group of LPR lanes owns one output row and walks its CSR row; each lane holds VEC
// consecutive columns per column chunk.  Gathered X rows come from L2/Infinity Cache at
// partition scale; the per-row (col, val) stream is a wave-uniform broadcast read.
template <int VEC> struct VecT;
template <> struct VecT<1> { using type = float; };
template <> struct VecT<4> { using type = float4; };

__device__ __forceinline__ void fma_vec(float& a, float w, float x) { a = fmaf(w, x, a); }
__device__ __forceinline__ void fma_vec(float4& a, float w, const float4& x) {
    a.x = fmaf(w, x.x, a.x); a.y = fmaf(w, x.y, a.y); a.z = fmaf(w, x.z, a.z); a.w = fmaf(w, x.w, a.w);
}
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// ---------------------------------------------------------------------------------------------
// Mates: mate[e] = id of the reverse edge (dst_e -> src_e) of edge e, or -1 -- the pairing behind the paired scorer forward
// (edge_score.hip MODE 3).  rev[e]: binary search for src_e in the out-row of dst_e (rows of a coalesced, row-sorted edge list are
// ascending in dst; on any other list the search may simply miss and the edge stays unmated, which only costs speed); a mate
// is kept only when the relation is mutual, so duplicates pair off one to one and self loops stay alone.
__global__ void __launch_bounds__(kT) edge_reverse(const int64_t* __restrict__ ei, int64_t n, const int* __restrict__ out_ptr,
                                                  const int* __restrict__ out_dst, const int* __restrict__ out_eid, int* __restrict__ rev) {
    const int64_t e = static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x;
    if (e >= n) return;
    const int s = static_cast<int>(ei[e]), d = static_cast<int>(ei[n + e]);
    int lo = out_ptr[d], hi = out_ptr[d + 1];
    const int end = hi;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (out_dst[mid] < s) lo = mid + 1; else hi = mid;
    }
    rev[e] = (lo < end && out_dst[lo] == s) ? out_eid[lo] : -1;
}
__global__ void __launch_bounds__(kT) edge_mate(const int* __restrict__ rev, int64_t n, int* __restrict__ mate) {
    const int64_t e = static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x;
    if (e >= n) return;
    const int m = rev[e];
    mate[e] = (m >= 0 && m != e && rev[m] == static_cast<int>(e)) ? m : -1;
}

template <int VEC, int LPR>
__global__ void __launch_bounds__(kT) spmm_csr(const float* __restrict__ X, int64_t N, int64_t D, const int* __restrict__ ptr,
                                              const int* __restrict__ col, const float* __restrict__ val,
                                              const float* __restrict__ diag, const float* __restrict__ bias, int act,
                                              float drop_scale, uint32_t drop_thresh, uint64_t seed, uint32_t site,
                                              const uint64_t* __restrict__ epoch, float* __restrict__ Y) {
    using V = typename VecT<VEC>::type;
    constexpr int RPB = kT / LPR;   // rows per block
    seed = fold_epoch(seed, epoch);
    const int sub = threadIdx.x % LPR;
    const int64_t i = static_cast<int64_t>(blockIdx.x) * RPB + threadIdx.x / LPR;
    if (i >= N) return;
    const int b = ptr[i], e = ptr[i + 1];
    const float dg = diag ? diag[i] : 0.f;
    for (int64_t c0 = static_cast<int64_t>(sub) * VEC; c0 < D; c0 += static_cast<int64_t>(LPR) * VEC) {
        V acc;
        if (VEC == 4) *reinterpret_cast<float4*>(&acc) = zero4(); else *reinterpret_cast<float*>(&acc) = 0.f;
        int k = b;
        for (; k + 4 <= e; k += 4) {          // 4 independent row gathers in flight
            const int j0 = col[k], j1 = col[k + 1], j2 = col[k + 2], j3 = col[k + 3];
            const float w0 = val[k], w1 = val[k + 1], w2 = val[k + 2], w3 = val[k + 3];
            const V x0 = *reinterpret_cast<const V*>(X + static_cast<int64_t>(j0) * D + c0);
            const V x1 = *reinterpret_cast<const V*>(X + static_cast<int64_t>(j1) * D + c0);
            const V x2 = *reinterpret_cast<const V*>(X + static_cast<int64_t>(j2) * D + c0);
            const V x3 = *reinterpret_cast<const V*>(X + static_cast<int64_t>(j3) * D + c0);
            fma_vec(acc, w0, x0); fma_vec(acc, w1, x1); fma_vec(acc, w2, x2); fma_vec(acc, w3, x3);
        }
        for (; k < e; ++k) {
            const V x0 = *reinterpret_cast<const V*>(X + static_cast<int64_t>(col[k]) * D + c0);
            fma_vec(acc, val[k], x0);
        }
        if (diag) {
            const V xi = *reinterpret_cast<const V*>(X + i * D + c0);
            fma_vec(acc, dg, xi);
        }
        float o[VEC];
        *reinterpret_cast<V*>(o) = acc;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            float y = o[v];
            if (bias) y += bias[c0 + v];
            if (act != SGS_ACT_NONE) y = fmaxf(y, 0.f);
            if (act == SGS_ACT_RELU_DROPOUT)
                y = dropout_keep_at(seed, site, static_cast<uint64_t>(i), static_cast<uint32_t>(c0 + v), drop_thresh)
                        ? y * drop_scale : 0.f;
            o[v] = y;
        }
        *reinterpret_cast<V*>(Y + i * D + c0) = *reinterpret_cast<V*>(o);
    }
}

// Small-N variant (METIS partitions: ~1k rows of ~100-1000 nnz): one wave per row leaves the chip
// mostly empty and hub rows serialise, so a 4-wave workgroup owns a row, each wave gathers a strided
// quarter of its nnz (8 rows in flight per wave) and the four partial sums are combined through LDS
// in a fixed order (deterministic).
// NW waves per row: 4, or 16 when rows are long (power-law partitions: the hub rows set the kernel's duration)
template <int VEC, int NW>
__global__ void __launch_bounds__(64 * NW) spmm_csr_rowblock(const float* __restrict__ X, int64_t N, int64_t D, const int* __restrict__ ptr,
                                                       const int* __restrict__ col, const float* __restrict__ val,
                                                       const float* __restrict__ diag, const float* __restrict__ bias, int act,
                                                       float drop_scale, uint32_t drop_thresh, uint64_t seed, uint32_t site,
                                                       const uint64_t* __restrict__ epoch, float* __restrict__ Y) {
    using V = typename VecT<VEC>::type;
    __shared__ float part[NW][64 * VEC];
    seed = fold_epoch(seed, epoch);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i = blockIdx.x;
    const int b = ptr[i], e = ptr[i + 1];
    const float dg = diag ? diag[i] : 0.f;
    const uint32_t rkey = dropout_row_key(seed, site, static_cast<uint64_t>(i));
    for (int64_t cbase = 0; cbase < D; cbase += 64 * VEC) {
        const int64_t c0 = cbase + static_cast<int64_t>(lane) * VEC;
        const bool in = c0 < D;
        float acc[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
        if (in) {
            int k = b + wave;
            for (; k + 7 * NW < e; k += 8 * NW) {    // 8 independent gathers (k, k+NW, ..., k+7NW)
                int j[8]; float w[8]; V x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { j[u] = col[k + NW * u]; w[u] = val[k + NW * u]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) x[u] = *reinterpret_cast<const V*>(X + static_cast<int64_t>(j[u]) * D + c0);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    float xv[VEC];
                    *reinterpret_cast<V*>(xv) = x[u];
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc[v] = fmaf(w[u], xv[v], acc[v]);
                }
            }
            for (; k < e; k += NW) {
                float xv[VEC];
                *reinterpret_cast<V*>(xv) = *reinterpret_cast<const V*>(X + static_cast<int64_t>(col[k]) * D + c0);
                const float w = val[k];
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc[v] = fmaf(w, xv[v], acc[v]);
            }
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) part[wave][lane * VEC + v] = acc[v];
        __syncthreads();
        // 64*VEC columns finished by the first 64*VEC threads
        const int t = threadIdx.x;
        if (t < 64 * VEC && cbase + t < D) {
            const int64_t c = cbase + t;
            float y = 0.f;
#pragma unroll
            for (int g = 0; g < NW; g += 4) y += (part[g][t] + part[g + 1][t]) + (part[g + 2][t] + part[g + 3][t]);
            if (diag) y = fmaf(dg, X[i * D + c], y);
            if (bias) y += bias[c];
            if (act != SGS_ACT_NONE) y = fmaxf(y, 0.f);
            if (act == SGS_ACT_RELU_DROPOUT) y = dropout_keep_col(rkey, static_cast<uint32_t>(c), drop_thresh) ? y * drop_scale : 0.f;
            Y[i * D + c] = y;
        }
        __syncthreads();
    }
}

// SDDMM over the CSR: g[eid[k]] = <A[i,:], B[col[k],:]> for k in row i ; gdiag[i] = <A[i,:], B[i,:]>.
template <int VEC, int LPR>
__global__ void __launch_bounds__(kT) sddmm_csr(const float* __restrict__ A, const float* __restrict__ B, int64_t N, int64_t D,
                                               const int* __restrict__ ptr, const int* __restrict__ col,
                                               const int* __restrict__ eid, float* __restrict__ g, float* __restrict__ gdiag) {
    using V = typename VecT<VEC>::type;
    constexpr int RPB = kT / LPR;
    const int sub = threadIdx.x % LPR;
    const int64_t i = static_cast<int64_t>(blockIdx.x) * RPB + threadIdx.x / LPR;
    const bool live = i < N;            // keep every lane in the shuffles
    const int b = live ? ptr[i] : 0, e = live ? ptr[i + 1] : 0;
    // rows of one LPR-group advance together; groups in a wave may have different trip counts,
    // so the loop bound is made wave-uniform (max over the wave) and dead lanes contribute 0.
    int trips = e - b + 1;             // +1: the diagonal term
    for (int o = 32; o > 0; o >>= 1) trips = max(trips, __shfl_xor(trips, o, 64));
    for (int t = 0; t < trips; ++t) {
        const int k = b + t;
        const bool is_edge = live && k < e;
        const bool is_diag = live && k == e;
        const int64_t j = is_edge ? col[k] : i;
        float acc = 0.f;
        if (is_edge || is_diag) {
            for (int64_t c0 = static_cast<int64_t>(sub) * VEC; c0 < D; c0 += static_cast<int64_t>(LPR) * VEC) {
                float a[VEC], x[VEC];
                *reinterpret_cast<V*>(a) = *reinterpret_cast<const V*>(A + i * D + c0);
                *reinterpret_cast<V*>(x) = *reinterpret_cast<const V*>(B + j * D + c0);
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc = fmaf(a[v], x[v], acc);
            }
        }
#pragma unroll
        for (int o = LPR >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (sub == 0) {
            if (is_edge) g[eid[k]] = acc;
            else if (is_diag && gdiag) gdiag[i] = acc;
        }
    }
}

// Small-N variant of sddmm_csr: a 4-wave workgroup per row, waves stride the row's entries, two
// independent dot products in flight per wave.
template <int VEC, int NW>
__global__ void __launch_bounds__(64 * NW) sddmm_csr_rowblock(const float* __restrict__ A, const float* __restrict__ B, int64_t N, int64_t D,
                                                        const int* __restrict__ ptr, const int* __restrict__ col,
                                                        const int* __restrict__ eid, float* __restrict__ g, float* __restrict__ gdiag) {
    using V = typename VecT<VEC>::type;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i = blockIdx.x;
    const int b = ptr[i], e = ptr[i + 1];
    const float* Ai = A + i * D;
    auto dot = [&](int64_t j) {
        float acc = 0.f;
        for (int64_t c0 = static_cast<int64_t>(lane) * VEC; c0 < D; c0 += 64 * VEC) {
            float a[VEC], x[VEC];
            *reinterpret_cast<V*>(a) = *reinterpret_cast<const V*>(Ai + c0);
            *reinterpret_cast<V*>(x) = *reinterpret_cast<const V*>(B + j * D + c0);
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc = fmaf(a[v], x[v], acc);
        }
        return acc;
    };
    if (D <= 64 * VEC) {
        // the usual case (a row fits one pass of the wave): row i's slice stays in registers, FOUR entries per step with their indices
        // loaded first and the four row gathers issued together (unconditional: entries past the row are clamped and not stored)
        const int64_t c0 = static_cast<int64_t>(lane) * VEC;
        const bool in = c0 < D;
        float a[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) a[v] = 0.f;
        if (in) *reinterpret_cast<V*>(a) = *reinterpret_cast<const V*>(Ai + c0);
        const int64_t cl = in ? c0 : 0;
        for (int k = b + wave; k < e; k += 4 * NW) {
            int jc[4], ev[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int kk = k + NW * u;
                const int kc = kk < e ? kk : e - 1;
                jc[u] = col[kc];
                ev[u] = eid[kc];
            }
            float x[4][VEC];
#pragma unroll
            for (int u = 0; u < 4; ++u) *reinterpret_cast<V*>(x[u]) = *reinterpret_cast<const V*>(B + static_cast<int64_t>(jc[u]) * D + cl);
            float d[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float acc = 0.f;
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc = fmaf(a[v], in ? x[u][v] : 0.f, acc);
                d[u] = acc;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) d[u] = wave_sum_hi_dpp(d[u]);       // (DPP path; the total lands in lane 63)
            if (lane == 63) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (k + NW * u < e) g[ev[u]] = d[u];
            }
        }
        if (wave == 0 && gdiag) {
            const float d0 = wave_sum_all(dot(i));
            if (lane == 0) gdiag[i] = d0;
        }
        return;
    }
    int k = b + wave;
    for (; k + NW < e; k += 2 * NW) {
        float d0 = dot(col[k]), d1 = dot(col[k + NW]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { d0 += __shfl_xor(d0, o, 64); d1 += __shfl_xor(d1, o, 64); }
        if (lane == 0) { g[eid[k]] = d0; g[eid[k + NW]] = d1; }
    }
    for (; k < e; k += NW) {
        const float d0 = wave_sum_all(dot(col[k]));
        if (lane == 0) g[eid[k]] = d0;
    }
    if (wave == 0 && gdiag) {
        const float d0 = wave_sum_all(dot(i));
        if (lane == 0) gdiag[i] = d0;
    }
}

// dZ = dY * act'(Y):  ReLU -> [Y > 0];  ReLU+dropout -> [Y > 0] / (1 - p)  (Y > 0 iff kept and positive).
__global__ void __launch_bounds__(kT) act_bwd(const float* __restrict__ dY, const float* __restrict__ Y, int64_t n, int act,
                                             float drop_scale, float* __restrict__ dZ) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x;
    if (i >= n) return;
    float g = dY[i];
    if (act != SGS_ACT_NONE) g = (Y[i] > 0.f) ? (act == SGS_ACT_RELU_DROPOUT ? g * drop_scale : g) : 0.f;
    dZ[i] = g;
}

// Column sums of a [N, D] matrix (bias / fc2 gradients), two fixed-order stages so the result is
// deterministic and the first stage fills the chip: block (cx, ry) sums kColRows rows of 64 columns.
constexpr int kColRows = 256;
// rows per first-stage workgroup: 256 for tall matrices, down to 32 for partition-sized ones so that the first stage
// still spreads over the chip (N = 1013: 32 row chunks x D/64 column groups instead of 4 x D/64)
inline int colsum_rows(int64_t N) { return N >= 16384 ? kColRows : (N >= 4096 ? 128 : 32); }
__global__ void __launch_bounds__(kT) colsum_partial(const float* __restrict__ A, int64_t N, int64_t D, int rows, float* __restrict__ part) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rgrp = threadIdx.x >> 6;
    const int64_t r0 = static_cast<int64_t>(blockIdx.y) * rows;
    const int64_t r1 = (r0 + rows < N) ? r0 + rows : N;
    float acc = 0.f;
    if (c < D)
        for (int64_t r = r0 + rgrp; r < r1; r += 4) acc += A[r * D + c];
    red[rgrp][threadIdx.x & 63] = acc;
    __syncthreads();
    if (rgrp == 0 && c < D)
        part[static_cast<int64_t>(blockIdx.y) * D + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
template <int RG>    // RG row groups of 64 columns per workgroup (RG * 64 threads): 4 for a handful of partials, 16 for tall inputs
__global__ void __launch_bounds__(RG * 64) colsum_final(const float* __restrict__ part, int64_t nchunk, int64_t D, float* __restrict__ out) {
    __shared__ float red[RG][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rgrp = threadIdx.x >> 6;
    float acc = 0.f;
    if (c < D)
        for (int64_t r = rgrp; r < nchunk; r += RG) acc += part[r * D + c];
    red[rgrp][threadIdx.x & 63] = acc;
    __syncthreads();
    if (rgrp == 0 && c < D) {
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < RG; g += 4) s += (red[g][threadIdx.x] + red[g + 1][threadIdx.x]) + (red[g + 2][threadIdx.x] + red[g + 3][threadIdx.x]);
        out[c] = s;
    }
}

// Partition-sized inputs (N <= kColSmallRows rows): the column sums in ONE launch -- a workgroup owns 16 columns, its 64 row lanes
// stride the rows and meet in LDS in a fixed tree (deterministic).  ACT: the activation backward rides along, dZ = dY * act'(Y) is
// written and summed in the same pass (a layer's backward then holds one launch where it held act_bwd + the two colsum stages).
constexpr int kColSmallRows = 2048;
template <bool ACT>
__global__ void __launch_bounds__(1024) colsum_small(const float* __restrict__ A, const float* __restrict__ Y, int64_t N, int64_t D, int act,
                                                    float drop_scale, float* __restrict__ dZ, float* __restrict__ out) {
    __shared__ float red[64][17];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int64_t c = static_cast<int64_t>(blockIdx.x) * 16 + cl;
    float acc = 0.f;
    if (c < D) {
#pragma unroll 16
        for (int64_t r = rl; r < N; r += 64) {        // (N ~ 1 000: all 16 loads of a thread in flight at once)
            float g = A[r * D + c];
            if (ACT) {
                if (act != SGS_ACT_NONE) g = (Y[r * D + c] > 0.f) ? (act == SGS_ACT_RELU_DROPOUT ? g * drop_scale : g) : 0.f;
                dZ[r * D + c] = g;
            }
            acc += g;
        }
    }
    red[rl][cl] = acc;
    __syncthreads();
#pragma unroll
    for (int h = 32; h >= 1; h >>= 1) {
        if (rl < h) red[rl][cl] += red[rl + h][cl];
        __syncthreads();
    }
    if (rl == 0 && c < D && out) out[c] = red[0][cl];
}

// D % 4 == 0: the same pass on 16-byte words -- a thread owns four columns and N/256 rows, so every load of a thread is in flight at
// once and the 16-column row segments stay whole 64-byte reads.
template <bool ACT>
__global__ void __launch_bounds__(1024) colsum_small_v4(const float* __restrict__ A, const float* __restrict__ Y, int64_t N, int64_t D, int act,
                                                       float drop_scale, float* __restrict__ dZ, float* __restrict__ out) {
    __shared__ float4 red[256][4];
    const int cg = threadIdx.x & 3, rl = threadIdx.x >> 2;
    const int64_t c = static_cast<int64_t>(blockIdx.x) * 16 + 4 * cg;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < D) {
#pragma unroll 8
        for (int64_t r = rl; r < N; r += 256) {
            float4 g = *reinterpret_cast<const float4*>(A + r * D + c);
            if (ACT) {
                if (act != SGS_ACT_NONE) {
                    const float4 y = *reinterpret_cast<const float4*>(Y + r * D + c);
                    const float sc = act == SGS_ACT_RELU_DROPOUT ? drop_scale : 1.f;
                    g.x = y.x > 0.f ? g.x * sc : 0.f; g.y = y.y > 0.f ? g.y * sc : 0.f;
                    g.z = y.z > 0.f ? g.z * sc : 0.f; g.w = y.w > 0.f ? g.w * sc : 0.f;
                }
                *reinterpret_cast<float4*>(dZ + r * D + c) = g;
            }
            acc.x += g.x; acc.y += g.y; acc.z += g.z; acc.w += g.w;
        }
    }
    red[rl][cg] = acc;
    __syncthreads();
#pragma unroll
    for (int h = 128; h >= 1; h >>= 1) {
        if (rl < h) {
            const float4 o = red[rl + h][cg];
            float4 m = red[rl][cg];
            m.x += o.x; m.y += o.y; m.z += o.z; m.w += o.w;
            red[rl][cg] = m;
        }
        __syncthreads();
    }
    if (rl == 0 && c < D && out) *reinterpret_cast<float4*>(out + c) = red[0][cg];
}

// A vector's sum (D = 1, e.g. d fc2.bias = sum of dz over the q active edges) in one workgroup: the two-stage kernels above would spend
// a 64-column workgroup on every row chunk of a one-column matrix.
constexpr int64_t kVecSumMax = int64_t(1) << 20;
__global__ void __launch_bounds__(1024) vecsum_small(const float* __restrict__ A, int64_t N, float* __restrict__ out) {
    __shared__ float red[16];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int64_t i = threadIdx.x;
    for (; i + 3 * 1024 < N; i += 4 * 1024) { a0 += A[i]; a1 += A[i + 1024]; a2 += A[i + 2048]; a3 += A[i + 3072]; }
    for (; i < N; i += 1024) a0 += A[i];
    const float r = block_sum((a0 + a1) + (a2 + a3), red);
    if (threadIdx.x == 0) out[0] = r;
}

inline int pick_lpr(int64_t D, int vec) {
    const int64_t need = (D + vec - 1) / vec;
    int lpr = 1;
    while (lpr < need && lpr < 64) lpr <<= 1;
    return lpr;
}

}  // namespace
}  // namespace sgs

using namespace sgs;

#define DISPATCH_VEC_LPR(KERNEL, vec, lpr, grid_rows, ...)                                                       \
    do {                                                                                                          \
        const int _rpb = kT / (lpr);                                                                              \
        const dim3 _g(static_cast<unsigned>(cdiv((grid_rows), _rpb))), _b(kT);                                    \
        if ((vec) == 4) {                                                                                         \
            switch (lpr) {                                                                                        \
                case 1: hipLaunchKernelGGL((KERNEL<4, 1>), _g, _b, 0, stream, __VA_ARGS__); break;                \
                case 2: hipLaunchKernelGGL((KERNEL<4, 2>), _g, _b, 0, stream, __VA_ARGS__); break;                \
                case 4: hipLaunchKernelGGL((KERNEL<4, 4>), _g, _b, 0, stream, __VA_ARGS__); break;                \
                case 8: hipLaunchKernelGGL((KERNEL<4, 8>), _g, _b, 0, stream, __VA_ARGS__); break;                \
                case 16: hipLaunchKernelGGL((KERNEL<4, 16>), _g, _b, 0, stream, __VA_ARGS__); break;              \
                case 32: hipLaunchKernelGGL((KERNEL<4, 32>), _g, _b, 0, stream, __VA_ARGS__); break;              \
                default: hipLaunchKernelGGL((KERNEL<4, 64>), _g, _b, 0, stream, __VA_ARGS__); break;              \
            }                                                                                                     \
        } else {                                                                                                  \
            switch (lpr) {                                                                                        \
                case 1: hipLaunchKernelGGL((KERNEL<1, 1>), _g, _b, 0, stream, __VA_ARGS__); break;                \
                case 2: hipLaunchKernelGGL((KERNEL<1, 2>), _g, _b, 0, stream, __VA_ARGS__); break;                \
                case 4: hipLaunchKernelGGL((KERNEL<1, 4>), _g, _b, 0, stream, __VA_ARGS__); break;                \
                case 8: hipLaunchKernelGGL((KERNEL<1, 8>), _g, _b, 0, stream, __VA_ARGS__); break;                \
                case 16: hipLaunchKernelGGL((KERNEL<1, 16>), _g, _b, 0, stream, __VA_ARGS__); break;              \
                case 32: hipLaunchKernelGGL((KERNEL<1, 32>), _g, _b, 0, stream, __VA_ARGS__); break;              \
                default: hipLaunchKernelGGL((KERNEL<1, 64>), _g, _b, 0, stream, __VA_ARGS__); break;              \
            }                                                                                                     \
        }                                                                                                         \
    } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" {

// large edge lists: two stable radix sorts instead of atomics + per-row sorts (csrc/graph_sort.hip)
constexpr int64_t kSortEdges = int64_t(1) << 22;

size_t sgs_graph_build_workspace_bytes(int64_t n_edges, int64_t N) {
    if (n_edges < 0) n_edges = 0;
    if (N < 0) N = 0;
    if (n_edges >= kSortEdges) return graph_sort_workspace_bytes(n_edges, N);
    return 4 * carve_bytes(N + 1, 4) + 2 * carve_bytes(n_edges + 1, 4) + 512;
}

int sgs_graph_build(const int64_t* edge_index, int64_t n_edges, int64_t N, int32_t* in_ptr, int32_t* in_src,
                    int32_t* in_eid, int32_t* out_ptr, int32_t* out_dst, int32_t* out_eid, int32_t* loop_eid, void* ws,
                    size_t ws_bytes, sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(n_edges >= 0 && N >= 0 && n_edges < (int64_t(1) << 31) && N < (int64_t(1) << 31), SGS_EINVAL,
                "sgs_graph_build: sizes out of range (n_edges=%lld N=%lld)", (long long)n_edges, (long long)N);
    if (N == 0) return SGS_OK;
    SGS_REQUIRE(in_ptr && out_ptr && loop_eid && (n_edges == 0 || (edge_index && in_src && in_eid && out_dst && out_eid)),
                SGS_EINVAL, "sgs_graph_build: null pointer");
    SGS_REQUIRE(ws && ws_bytes >= sgs_graph_build_workspace_bytes(n_edges, N), SGS_EWORKSPACE,
                "sgs_graph_build: workspace too small");
    if (n_edges >= kSortEdges)
        return graph_build_by_sort(edge_index, n_edges, N, in_ptr, in_src, in_eid, out_ptr, out_dst, out_eid, loop_eid, ws, ws_bytes, stream);
    Carver cv(ws);
    int* cnt_in = cv.take<int>(N + 1);
    int* cnt_out = cv.take<int>(N + 1);
    int* cur_in = cv.take<int>(N + 1);
    int* cur_out = cv.take<int>(N + 1);
    int* tmp_in = cv.take<int>(n_edges + 1);
    int* tmp_out = cv.take<int>(n_edges + 1);
    // cnt_in + cnt_out are adjacent -> 0; loop_eid -> -1 (one kernel launch, not memset nodes: see zero_async)
    if (int rc = fill2_async(cnt_in, 2 * carve_bytes(N + 1, 4), 0u, loop_eid, static_cast<size_t>(N) * 4, 0xFFFFFFFFu, stream)) return rc;
    const bool small_n = N <= kLdsNodes;
    if (n_edges > 0) {
        if (small_n)
            hipLaunchKernelGGL(count_degrees_lds, dim3(cdiv(n_edges, kEdgesPerBlock)), dim3(kT), 0, stream, edge_index, n_edges,
                               static_cast<int>(N), cnt_in, cnt_out, loop_eid);
        else
            hipLaunchKernelGGL(count_degrees, dim3(cdiv(n_edges, kT)), dim3(kT), 0, stream, edge_index, n_edges, cnt_in, cnt_out,
                               loop_eid);
    }
    // (a "last workgroup scans" variant of the count kernel was tried and is slower on MI355X: the device-scope fences it
    //  needs write back / invalidate the per-XCD L2s)
    hipLaunchKernelGGL(scan_counts, dim3(2), dim3(1024), 0, stream, cnt_in, cnt_out, N, in_ptr, out_ptr, cur_in, cur_out);
    if (n_edges > 0) {
        if (small_n)
            hipLaunchKernelGGL(fill_rows_lds, dim3(cdiv(n_edges, kEdgesPerBlock)), dim3(kT), 0, stream, edge_index, n_edges,
                               static_cast<int>(N), cur_in, cur_out, tmp_in, tmp_out);
        else
            hipLaunchKernelGGL(fill_rows, dim3(cdiv(n_edges, kT)), dim3(kT), 0, stream, edge_index, n_edges, cur_in, cur_out,
                               tmp_in, tmp_out);
        const int64_t n_wave_blocks = cdiv(2 * N * 64, kT);
        hipLaunchKernelGGL(sort_rows, dim3(static_cast<unsigned>(n_wave_blocks + 2 * N)), dim3(kT), 0, stream, n_wave_blocks, in_ptr, out_ptr, N,
                           tmp_in, tmp_out, edge_index, n_edges, in_eid, in_src, out_eid, out_dst);
    }
    SGS_LAUNCH_OK();
    return SGS_OK;
}

size_t sgs_graph_build_src_sorted_workspace_bytes(int64_t n_edges, int64_t N) {
    return graph_src_sorted_workspace_bytes(n_edges < 0 ? 0 : n_edges, N < 0 ? 0 : N);
}

int sgs_graph_build_src_sorted(const int64_t* edge_index, int64_t n_edges, int64_t N, int32_t* in_ptr, int32_t* in_src, int32_t* in_eid,
                               int32_t* out_ptr, int32_t* out_dst, int32_t* out_eid, int32_t* loop_eid, int32_t* unsorted, void* ws,
                               size_t ws_bytes, sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(n_edges >= 0 && N >= 0 && n_edges < (int64_t(1) << 31) && N < (int64_t(1) << 31), SGS_EINVAL,
                "sgs_graph_build_src_sorted: sizes out of range (n_edges=%lld N=%lld)", (long long)n_edges, (long long)N);
    if (N == 0) return SGS_OK;
    SGS_REQUIRE(in_ptr && out_ptr && loop_eid && (n_edges == 0 || (edge_index && in_src && in_eid && out_dst && out_eid)), SGS_EINVAL,
                "sgs_graph_build_src_sorted: null pointer");
    SGS_REQUIRE(ws && (reinterpret_cast<uintptr_t>(ws) & 255) == 0, SGS_EINVAL, "sgs_graph_build_src_sorted: workspace must be 256-B aligned");
    return graph_build_src_sorted(edge_index, n_edges, N, in_ptr, in_src, in_eid, out_ptr, out_dst, out_eid, loop_eid, unsorted, ws, ws_bytes, stream);
}

size_t sgs_graph_filter_workspace_bytes(int64_t E_parent, int64_t N) {
    if (E_parent < 0) E_parent = 0;
    if (N < 0) N = 0;
    return 4 * carve_bytes(N + 1, 4) + carve_bytes(E_parent + 1, 4) + carve_bytes(E_parent / 32 + 2, 4) + 256;
}

int sgs_graph_filter(const int32_t* pin_ptr, const int32_t* pin_src, const int32_t* pin_eid, const int32_t* pout_ptr,
                     const int32_t* pout_dst, const int32_t* pout_eid, int64_t E_parent, int64_t N, const uint8_t* mask,
                     const int64_t* sampled_eid, int64_t q, int32_t* in_ptr, int32_t* in_src, int32_t* in_eid, int32_t* out_ptr,
                     int32_t* out_dst, int32_t* out_eid, int32_t* loop_eid, void* ws, size_t ws_bytes, sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(E_parent >= 0 && N >= 0 && q >= 0 && q <= E_parent && E_parent < (int64_t(1) << 31) && N < (int64_t(1) << 30), SGS_EINVAL,
                "sgs_graph_filter: bad sizes");
    if (N == 0) return SGS_OK;
    SGS_REQUIRE(pin_ptr && pout_ptr && in_ptr && out_ptr && loop_eid && (E_parent == 0 || (pin_src && pin_eid && pout_dst && pout_eid && mask)) &&
                    (q == 0 || (sampled_eid && in_src && in_eid && out_dst && out_eid)), SGS_EINVAL, "sgs_graph_filter: null pointer");
    SGS_REQUIRE(ws && ws_bytes >= sgs_graph_filter_workspace_bytes(E_parent, N), SGS_EWORKSPACE, "sgs_graph_filter: workspace too small");
    Carver cv(ws);
    int* cnt_in = cv.take<int>(N + 1);
    int* cnt_out = cv.take<int>(N + 1);
    int* cur_in = cv.take<int>(N + 1);       // scan_counts also initialises fill cursors; unused here
    int* cur_out = cv.take<int>(N + 1);
    int* pos = cv.take<int>(E_parent + 1);
    uint32_t* bits = cv.take<uint32_t>(E_parent / 32 + 2);
    const int64_t n_row_blocks = cdiv(2 * N * 64, kT);
    const dim3 g1(static_cast<unsigned>(n_row_blocks + cdiv(q, kT))), g3(static_cast<unsigned>(n_row_blocks)), blk(kT);
    if (E_parent >= (int64_t(1) << 23)) {
        hipLaunchKernelGGL(pack_mask_bits, dim3(static_cast<unsigned>(cdiv(cdiv(E_parent, 32), kT))), blk, 0, stream, mask, E_parent, bits);
        const uint8_t* bm = reinterpret_cast<const uint8_t*>(bits);
        hipLaunchKernelGGL(filter_count_scatter<true>, g1, blk, 0, stream, pin_ptr, pin_eid, pout_ptr, pout_eid, N, bm, sampled_eid, q, n_row_blocks,
                           cnt_in, cnt_out, pos);
        hipLaunchKernelGGL(scan_counts, dim3(2), dim3(1024), 0, stream, cnt_in, cnt_out, N, in_ptr, out_ptr, cur_in, cur_out);
        hipLaunchKernelGGL(filter_fill<true>, g3, blk, 0, stream, pin_ptr, pin_src, pin_eid, pout_ptr, pout_dst, pout_eid, N, bm, pos, in_ptr, in_src,
                           in_eid, out_ptr, out_dst, out_eid, loop_eid);
    } else if (N <= kSelfScanRows) {
        hipLaunchKernelGGL(filter_count_scatter<false>, g1, blk, 0, stream, pin_ptr, pin_eid, pout_ptr, pout_eid, N, mask, sampled_eid, q,
                           n_row_blocks, cnt_in, cnt_out, pos);
        hipLaunchKernelGGL((filter_fill<false, true>), g3, blk, 0, stream, pin_ptr, pin_src, pin_eid, pout_ptr, pout_dst, pout_eid, N, mask, pos,
                           in_ptr, in_src, in_eid, out_ptr, out_dst, out_eid, loop_eid, static_cast<const int*>(cnt_in),
                           static_cast<const int*>(cnt_out));
    } else {
        hipLaunchKernelGGL(filter_count_scatter<false>, g1, blk, 0, stream, pin_ptr, pin_eid, pout_ptr, pout_eid, N, mask, sampled_eid, q,
                           n_row_blocks, cnt_in, cnt_out, pos);
        hipLaunchKernelGGL(scan_counts, dim3(2), dim3(1024), 0, stream, cnt_in, cnt_out, N, in_ptr, out_ptr, cur_in, cur_out);
        hipLaunchKernelGGL(filter_fill<false>, g3, blk, 0, stream, pin_ptr, pin_src, pin_eid, pout_ptr, pout_dst, pout_eid, N, mask, pos, in_ptr,
                           in_src, in_eid, out_ptr, out_dst, out_eid, loop_eid);
    }
    SGS_LAUNCH_OK();
    return SGS_OK;
}

size_t sgs_edge_mates_workspace_bytes(int64_t n_edges) { return carve_bytes(n_edges < 0 ? 0 : n_edges, 4) + 256; }

int sgs_edge_mates(const int64_t* edge_index, int64_t n_edges, int64_t N, const int32_t* out_ptr, const int32_t* out_dst,
                   const int32_t* out_eid, int32_t* mate, void* ws, size_t ws_bytes, sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(n_edges >= 0 && N >= 0 && n_edges < (int64_t(1) << 31), SGS_EINVAL, "sgs_edge_mates: bad sizes");
    if (n_edges == 0) return SGS_OK;
    SGS_REQUIRE(edge_index && out_ptr && out_dst && out_eid && mate, SGS_EINVAL, "sgs_edge_mates: null pointer");
    SGS_REQUIRE(ws && ws_bytes >= sgs_edge_mates_workspace_bytes(n_edges), SGS_EWORKSPACE, "sgs_edge_mates: workspace too small");
    Carver cv(ws);
    int* rev = cv.take<int>(n_edges);
    const dim3 g(static_cast<unsigned>(cdiv(n_edges, kT))), blk(kT);
    hipLaunchKernelGGL(edge_reverse, g, blk, 0, stream, edge_index, n_edges, out_ptr, out_dst, out_eid, rev);
    hipLaunchKernelGGL(edge_mate, g, blk, 0, stream, rev, n_edges, mate);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

int sgs_gcn_norm_fwd(const float* w, int64_t n_edges, int64_t N, const int32_t* in_ptr, const int32_t* in_src,
                     const int32_t* in_eid, const int32_t* out_ptr, const int32_t* out_dst, const int32_t* out_eid,
                     const int32_t* loop_eid, float* dis, float* loopw, float* what_in, float* what_out, float* what_loop,
                     sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(N >= 0 && n_edges >= 0, SGS_EINVAL, "sgs_gcn_norm_fwd: bad sizes");
    if (N == 0) return SGS_OK;
    SGS_REQUIRE(in_ptr && out_ptr && loop_eid && dis && loopw && what_loop, SGS_EINVAL, "sgs_gcn_norm_fwd: null pointer");
    hipLaunchKernelGGL(norm_deg, dim3(cdiv(N * 64, kT)), dim3(kT), 0, stream, w, N, in_ptr, in_src, in_eid, loop_eid, dis, loopw);
    hipLaunchKernelGGL(norm_weights, dim3(cdiv(2 * N * 64, kT)), dim3(kT), 0, stream, w, N, n_edges, in_ptr, in_src, in_eid,
                       out_ptr, out_dst, out_eid, dis, loopw, what_in, what_out, what_loop);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

int sgs_gcn_degree_partial(const float* w, int64_t n_edges, int64_t N, const int32_t* in_ptr, const int32_t* in_src,
                           const int32_t* in_eid, float* degpart, sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(N >= 0 && n_edges >= 0, SGS_EINVAL, "sgs_gcn_degree_partial: bad sizes");
    if (N == 0) return SGS_OK;
    SGS_REQUIRE(in_ptr && degpart, SGS_EINVAL, "sgs_gcn_degree_partial: null pointer");
    hipLaunchKernelGGL(deg_partial, dim3(cdiv(N * 64, kT)), dim3(kT), 0, stream, w, N, in_ptr, in_src, in_eid, degpart);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

int sgs_gcn_norm_from_degree(const float* w, const float* degsum, int64_t n_edges, int64_t N, const int32_t* in_ptr,
                             const int32_t* in_src, const int32_t* in_eid, const int32_t* out_ptr, const int32_t* out_dst,
                             const int32_t* out_eid, float* dis, float* loopw, float* what_in, float* what_out,
                             float* what_loop, sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(N >= 0 && n_edges >= 0, SGS_EINVAL, "sgs_gcn_norm_from_degree: bad sizes");
    if (N == 0) return SGS_OK;
    SGS_REQUIRE(degsum && in_ptr && out_ptr && dis && loopw && what_loop, SGS_EINVAL, "sgs_gcn_norm_from_degree: null pointer");
    hipLaunchKernelGGL(dis_from_degree, dim3(cdiv(N, kT)), dim3(kT), 0, stream, degsum, N, dis, loopw);
    hipLaunchKernelGGL(norm_weights, dim3(cdiv(2 * N * 64, kT)), dim3(kT), 0, stream, w, N, n_edges, in_ptr, in_src, in_eid,
                       out_ptr, out_dst, out_eid, dis, loopw, what_in, what_out, what_loop);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

int sgs_bias_act(const float* X, const float* bias, int64_t N, int64_t D, int act, float p_drop, uint64_t seed, uint32_t site,
                 float* Y, sgs_stream_t stream_) {
    return sgs_bias_act_rows(X, bias, N, D, 0, act, p_drop, seed, site, Y, stream_);
}

int sgs_bias_act_rows(const float* X, const float* bias, int64_t N, int64_t D, int64_t row_offset, int act, float p_drop, uint64_t seed,
                      uint32_t site, float* Y, sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(N >= 0 && D >= 0 && D < (int64_t(1) << 32) && act >= SGS_ACT_NONE && act <= SGS_ACT_RELU_DROPOUT && p_drop >= 0.f &&
                    p_drop < 1.f, SGS_EINVAL, "sgs_bias_act: bad arguments");
    if (N * D == 0) return SGS_OK;
    SGS_REQUIRE(X && Y, SGS_EINVAL, "sgs_bias_act: null pointer");
    if (act == SGS_ACT_RELU_DROPOUT && p_drop == 0.f) act = SGS_ACT_RELU;
    hipLaunchKernelGGL(bias_act, dim3(cdiv(N * D, kT)), dim3(kT), 0, stream, X, bias, N, D, act, 1.0f / (1.0f - p_drop),
                       dropout_thresh(p_drop), seed, site, epoch_ptr(), Y, row_offset);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

int sgs_mean_weights(int64_t n_edges, int64_t N, const int32_t* in_ptr, const int32_t* out_ptr, const int32_t* out_dst,
                     float* what_in, float* what_out, sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(N >= 0 && n_edges >= 0, SGS_EINVAL, "sgs_mean_weights: bad sizes");
    if (N == 0 || n_edges == 0) return SGS_OK;
    SGS_REQUIRE(in_ptr && out_ptr && out_dst && what_in && what_out, SGS_EINVAL, "sgs_mean_weights: null pointer");
    hipLaunchKernelGGL(mean_weights, dim3(cdiv(2 * N * 64, kT)), dim3(kT), 0, stream, N, in_ptr, out_ptr, out_dst, what_in, what_out);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

int sgs_degree_prior_logits(const int64_t* edge_index, int64_t E, int64_t N, const int32_t* in_ptr, const int32_t* out_ptr,
                            float* logits, sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(N >= 0 && E >= 0, SGS_EINVAL, "sgs_degree_prior_logits: bad sizes");
    if (E == 0) return SGS_OK;
    SGS_REQUIRE(edge_index && in_ptr && out_ptr && logits, SGS_EINVAL, "sgs_degree_prior_logits: null pointer");
    const float scale = static_cast<float>(1.0 / sqrt(static_cast<double>(E)));          // len(prob) ** -0.5
    hipLaunchKernelGGL(degree_prior_logits, dim3(cdiv(E, kT)), dim3(kT), 0, stream, edge_index, E, in_ptr, out_ptr, scale, logits);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

size_t sgs_gcn_norm_bwd_workspace_bytes(int64_t N) { return carve_bytes(N < 0 ? 0 : N, 4) + 256; }

int sgs_gcn_norm_bwd(const float* w, const float* gw_hat, const float* gloop, int64_t n_edges, int64_t N,
                     const float* dis, const float* loopw, const int32_t* in_ptr, const int32_t* in_src,
                     const int32_t* in_eid, const int32_t* out_ptr, const int32_t* out_dst, const int32_t* out_eid,
                     const int32_t* loop_eid, const int64_t* edge_index, float* dw, void* ws, size_t ws_bytes,
                     sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(N >= 0 && n_edges >= 0, SGS_EINVAL, "sgs_gcn_norm_bwd: bad sizes");
    if (N == 0 || n_edges == 0) return SGS_OK;
    SGS_REQUIRE(w && gw_hat && gloop && dis && loopw && dw && edge_index, SGS_EINVAL, "sgs_gcn_norm_bwd: null pointer");
    SGS_REQUIRE(ws && ws_bytes >= sgs_gcn_norm_bwd_workspace_bytes(N), SGS_EWORKSPACE, "sgs_gcn_norm_bwd: workspace too small");
    Carver cv(ws);
    float* Hn = cv.take<float>(N);
    hipLaunchKernelGGL(norm_bwd_node, dim3(cdiv(N * 64, kT)), dim3(kT), 0, stream, w, gw_hat, gloop, N, in_ptr, in_src, in_eid,
                       out_ptr, out_dst, out_eid, dis, loopw, Hn);
    hipLaunchKernelGGL(norm_bwd_edge, dim3(cdiv(n_edges, kT)), dim3(kT), 0, stream, gw_hat, gloop, edge_index, n_edges, loop_eid,
                       dis, Hn, dw);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

/* The same with up to two upstream gradients summed on read (gw_hat2 / gloop2: the second GCN layer over the same normalisation) and
 * another consumer's d w added on the way out (dw_add, may be dw itself: in-place accumulation) -- the adds autograd would launch. */
int sgs_gcn_norm_bwd_sum(const float* w, const float* gw_hat, const float* gloop, const float* gw_hat2, const float* gloop2, const float* dw_add,
                         int64_t n_edges, int64_t N, const float* dis, const float* loopw, const int32_t* in_ptr, const int32_t* in_src,
                         const int32_t* in_eid, const int32_t* out_ptr, const int32_t* out_dst, const int32_t* out_eid,
                         const int32_t* loop_eid, const int64_t* edge_index, float* dw, void* ws, size_t ws_bytes, sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(N >= 0 && n_edges >= 0, SGS_EINVAL, "sgs_gcn_norm_bwd_sum: bad sizes");
    if (N == 0 || n_edges == 0) return SGS_OK;
    SGS_REQUIRE(w && gw_hat && gloop && dis && loopw && dw && edge_index && ((gw_hat2 != nullptr) == (gloop2 != nullptr)), SGS_EINVAL,
                "sgs_gcn_norm_bwd_sum: null pointer");
    SGS_REQUIRE(ws && ws_bytes >= sgs_gcn_norm_bwd_workspace_bytes(N), SGS_EWORKSPACE, "sgs_gcn_norm_bwd_sum: workspace too small");
    Carver cv(ws);
    float* Hn = cv.take<float>(N);
    hipLaunchKernelGGL(norm_bwd_node, dim3(cdiv(N * 64, kT)), dim3(kT), 0, stream, w, gw_hat, gloop, N, in_ptr, in_src, in_eid,
                       out_ptr, out_dst, out_eid, dis, loopw, Hn, gw_hat2, gloop2);
    hipLaunchKernelGGL(norm_bwd_edge, dim3(cdiv(n_edges, kT)), dim3(kT), 0, stream, gw_hat, gloop, edge_index, n_edges, loop_eid,
                       dis, Hn, dw, gw_hat2, gloop2, dw_add);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

/* Edge-sharded gcn_norm backward: the per-node term Hn is linear in its edge contributions, so each rank
 * computes its partial (gloop only on the rank that owns the self-loop term), the host all-reduces Hn, and
 * the per-edge pass finishes locally. */
int sgs_gcn_norm_bwd_node(const float* w, const float* gw_hat, const float* gloop, int64_t n_edges, int64_t N, const float* dis,
                          const float* loopw, const int32_t* in_ptr, const int32_t* in_src, const int32_t* in_eid,
                          const int32_t* out_ptr, const int32_t* out_dst, const int32_t* out_eid, float* Hn,
                          sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(N >= 0 && n_edges >= 0, SGS_EINVAL, "sgs_gcn_norm_bwd_node: bad sizes");
    if (N == 0) return SGS_OK;
    SGS_REQUIRE(w && gw_hat && gloop && dis && loopw && Hn, SGS_EINVAL, "sgs_gcn_norm_bwd_node: null pointer");
    hipLaunchKernelGGL(norm_bwd_node, dim3(cdiv(N * 64, kT)), dim3(kT), 0, stream, w, gw_hat, gloop, N, in_ptr, in_src, in_eid, out_ptr,
                       out_dst, out_eid, dis, loopw, Hn);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

int sgs_gcn_norm_bwd_edge(const float* gw_hat, const float* gloop, int64_t n_edges, int64_t N, const float* dis,
                          const int32_t* loop_eid, const int64_t* edge_index, const float* Hn, float* dw, sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(N >= 0 && n_edges >= 0, SGS_EINVAL, "sgs_gcn_norm_bwd_edge: bad sizes");
    if (n_edges == 0) return SGS_OK;
    SGS_REQUIRE(gw_hat && gloop && dis && loop_eid && edge_index && Hn && dw, SGS_EINVAL, "sgs_gcn_norm_bwd_edge: null pointer");
    hipLaunchKernelGGL(norm_bwd_edge, dim3(cdiv(n_edges, kT)), dim3(kT), 0, stream, gw_hat, gloop, edge_index, n_edges, loop_eid, dis, Hn, dw);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

int sgs_spmm_csr(const float* X, int64_t N, int64_t D, int64_t nnz, const int32_t* ptr, const int32_t* col, const float* val,
                 const float* diag, const float* bias, int act, float p_drop, uint64_t seed, uint32_t site, float* Y,
                 sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(N >= 0 && D >= 0, SGS_EINVAL, "sgs_spmm_csr: bad sizes");
    SGS_REQUIRE(act >= SGS_ACT_NONE && act <= SGS_ACT_RELU_DROPOUT && p_drop >= 0.f && p_drop < 1.f, SGS_EINVAL,
                "sgs_spmm_csr: bad activation / dropout");
    if (N == 0 || D == 0) return SGS_OK;
    SGS_REQUIRE(X && ptr && Y && X != Y, SGS_EINVAL, "sgs_spmm_csr: null or aliased pointer");
    const int vec = (D % 4 == 0 && aligned16(X) && aligned16(Y)) ? 4 : 1;
    const int lpr = pick_lpr(D, vec);
    const float scale = 1.0f / (1.0f - p_drop);
    const uint32_t th = dropout_thresh(p_drop);
    if (act == SGS_ACT_RELU_DROPOUT && p_drop == 0.f) act = SGS_ACT_RELU;
    if (N <= 65536 && nnz >= 16 * N) {        // few, long rows: a workgroup per row
        const bool wide = nnz >= 256 * N;      // very long rows only: at ~100 entries per row the 16-wave form measured 5 % slower here
        const dim3 g_(static_cast<unsigned>(N));
        if (vec == 4 && wide)
            hipLaunchKernelGGL((spmm_csr_rowblock<4, 16>), g_, dim3(1024), 0, stream, X, N, D, ptr, col, val, diag, bias, act, scale, th, seed,
                               site, epoch_ptr(), Y);
        else if (vec == 4)
            hipLaunchKernelGGL((spmm_csr_rowblock<4, 4>), g_, dim3(kT), 0, stream, X, N, D, ptr, col, val, diag, bias, act, scale, th, seed,
                               site, epoch_ptr(), Y);
        else if (wide)
            hipLaunchKernelGGL((spmm_csr_rowblock<1, 16>), g_, dim3(1024), 0, stream, X, N, D, ptr, col, val, diag, bias, act, scale, th, seed,
                               site, epoch_ptr(), Y);
        else
            hipLaunchKernelGGL((spmm_csr_rowblock<1, 4>), g_, dim3(kT), 0, stream, X, N, D, ptr, col, val, diag, bias, act, scale, th, seed,
                               site, epoch_ptr(), Y);
    } else {
        DISPATCH_VEC_LPR(spmm_csr, vec, lpr, N, X, N, D, ptr, col, val, diag, bias, act, scale, th, seed, site, epoch_ptr(), Y);
    }
    SGS_LAUNCH_OK();
    return SGS_OK;
}

int sgs_sddmm_csr(const float* A, const float* B, int64_t N, int64_t D, int64_t nnz, const int32_t* ptr, const int32_t* col,
                  const int32_t* eid, float* g, float* gdiag, sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(N >= 0 && D >= 0, SGS_EINVAL, "sgs_sddmm_csr: bad sizes");
    if (N == 0) return SGS_OK;
    SGS_REQUIRE(A && B && ptr, SGS_EINVAL, "sgs_sddmm_csr: null pointer");
    const int vec = (D % 4 == 0 && aligned16(A) && aligned16(B)) ? 4 : 1;
    const int lpr = pick_lpr(D, vec);
    if (nnz >= 16 * N) {                      // long rows (partitions; whole graphs of average degree >= 16): a workgroup per row
        const bool wide = nnz >= 64 * N;
        const dim3 g_(static_cast<unsigned>(N));
        if (vec == 4 && wide) hipLaunchKernelGGL((sddmm_csr_rowblock<4, 16>), g_, dim3(1024), 0, stream, A, B, N, D, ptr, col, eid, g, gdiag);
        else if (vec == 4)    hipLaunchKernelGGL((sddmm_csr_rowblock<4, 4>), g_, dim3(kT), 0, stream, A, B, N, D, ptr, col, eid, g, gdiag);
        else if (wide)        hipLaunchKernelGGL((sddmm_csr_rowblock<1, 16>), g_, dim3(1024), 0, stream, A, B, N, D, ptr, col, eid, g, gdiag);
        else                  hipLaunchKernelGGL((sddmm_csr_rowblock<1, 4>), g_, dim3(kT), 0, stream, A, B, N, D, ptr, col, eid, g, gdiag);
    } else {
        DISPATCH_VEC_LPR(sddmm_csr, vec, lpr, N, A, B, N, D, ptr, col, eid, g, gdiag);
    }
    SGS_LAUNCH_OK();
    return SGS_OK;
}

int sgs_act_bwd(const float* dY, const float* Y, int64_t n, int act, float p_drop, float* dZ, sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(n >= 0 && p_drop >= 0.f && p_drop < 1.f, SGS_EINVAL, "sgs_act_bwd: bad arguments");
    if (n == 0) return SGS_OK;
    SGS_REQUIRE(dY && dZ && (act == SGS_ACT_NONE || Y), SGS_EINVAL, "sgs_act_bwd: null pointer");
    if (act == SGS_ACT_RELU_DROPOUT && p_drop == 0.f) act = SGS_ACT_RELU;
    hipLaunchKernelGGL(act_bwd, dim3(cdiv(n, kT)), dim3(kT), 0, stream, dY, Y, n, act, 1.0f / (1.0f - p_drop), dZ);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

size_t sgs_colsum_workspace_bytes(int64_t N, int64_t D) {
    if (N < 0) N = 0;
    if (D < 0) D = 0;
    return carve_bytes(static_cast<size_t>(cdiv(N, colsum_rows(N)) + 1) * D, 4) + 256;
}

int sgs_colsum(const float* A, int64_t N, int64_t D, float* out, void* ws, size_t ws_bytes, sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(N >= 0 && D >= 0, SGS_EINVAL, "sgs_colsum: bad sizes");
    if (D == 0) return SGS_OK;
    SGS_REQUIRE(out && (N == 0 || A), SGS_EINVAL, "sgs_colsum: null pointer");
    SGS_REQUIRE(ws && ws_bytes >= sgs_colsum_workspace_bytes(N, D), SGS_EWORKSPACE, "sgs_colsum: workspace too small");
    if (N > 0 && N <= kColSmallRows) {
        // (round 1 tried a single launch with 64-column workgroups, N/16 rows per thread: slower than the two stages by 12 us per
        //  backward; with 16-column workgroups a thread walks N/64 rows and all of its loads are in flight together)
        if (D % 4 == 0)
            hipLaunchKernelGGL(colsum_small_v4<false>, dim3(cdiv(D, 16)), dim3(1024), 0, stream, A, static_cast<const float*>(nullptr), N, D, SGS_ACT_NONE,
                               1.f, static_cast<float*>(nullptr), out);
        else
            hipLaunchKernelGGL(colsum_small<false>, dim3(cdiv(D, 16)), dim3(1024), 0, stream, A, static_cast<const float*>(nullptr), N, D, SGS_ACT_NONE, 1.f,
                               static_cast<float*>(nullptr), out);
        SGS_LAUNCH_OK();
        return SGS_OK;
    }
    if (D == 1 && N > 0 && N <= kVecSumMax) {
        hipLaunchKernelGGL(vecsum_small, dim3(1), dim3(1024), 0, stream, A, N, out);
        SGS_LAUNCH_OK();
        return SGS_OK;
    }
    Carver cv(ws);
    const int rows = colsum_rows(N);
    const int64_t nchunk = cdiv(N, rows);
    float* part = cv.take<float>(static_cast<size_t>(nchunk + 1) * D);
    if (nchunk > 0)
        hipLaunchKernelGGL(colsum_partial, dim3(cdiv(D, 64), nchunk), dim3(kT), 0, stream, A, N, D, rows, part);
    if (nchunk > 64) hipLaunchKernelGGL(colsum_final<16>, dim3(cdiv(D, 64)), dim3(1024), 0, stream, part, nchunk, D, out);
    else             hipLaunchKernelGGL(colsum_final<4>, dim3(cdiv(D, 64)), dim3(kT), 0, stream, part, nchunk, D, out);
    SGS_LAUNCH_OK();
    return SGS_OK;
}


int sgs_act_bwd_colsum(const float* dY, const float* Y, int64_t N, int64_t D, int act, float p_drop, float* dZ, float* colsum, void* ws,
                       size_t ws_bytes, sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(N >= 0 && D >= 0 && p_drop >= 0.f && p_drop < 1.f, SGS_EINVAL, "sgs_act_bwd_colsum: bad arguments");
    if (D == 0) return SGS_OK;
    SGS_REQUIRE(dZ && colsum && (N == 0 || (dY && (act == SGS_ACT_NONE || Y))), SGS_EINVAL, "sgs_act_bwd_colsum: null pointer");
    if (N > 0 && N <= kColSmallRows) {
        if (D % 4 == 0) hipLaunchKernelGGL(colsum_small_v4<true>, dim3(cdiv(D, 16)), dim3(1024), 0, stream, dY, Y, N, D, act, 1.0f / (1.0f - p_drop), dZ, colsum);
        else            hipLaunchKernelGGL(colsum_small<true>, dim3(cdiv(D, 16)), dim3(1024), 0, stream, dY, Y, N, D, act, 1.0f / (1.0f - p_drop), dZ, colsum);
        SGS_LAUNCH_OK();
        return SGS_OK;
    }
    if (int rc = sgs_act_bwd(dY, Y, N * D, act, p_drop, dZ, stream_)) return rc;
    return sgs_colsum(dZ, N, D, colsum, ws, ws_bytes, stream_);
}

}  // extern "C"
