/*
 * sgs_hip.h -- C ABI of libsgs_hip.so, the MI355X (gfx950) implementation of the SGS-GNN
 * hot path: edge scoring -> exponential-race top-q edge sampling -> weighted sparse GCN
 * forward/backward (+ gate and losses).
 *
 * The reference (anonymousauthors001/SGS-GNN) has no native/FFI layer: its boundary is the
 * Python call sites of training_hybrid.py / sampling.py / model.py.  Each entry point below
 * names the reference lines whose device work it replaces; INTEGRATION.md shows the ctypes
 * binding a maintainer would add on the reference side.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless marked "host"; tensors are dense row-major;
 *   - `edge_index` is PyG's int64 [2,E] layout: src = edge_index, dst = edge_index + E;
 *   - inputs are borrowed, outputs are caller-allocated; no entry point allocates, frees,
 *     synchronises the device or keeps global state -> all are HIP-graph capturable;
 *   - scratch memory comes from the caller: query `*_workspace_bytes`, pass `ws, ws_bytes`;
 *   - work is enqueued on `stream` (a hipStream_t cast to void*; NULL = default stream);
 *   - return value: SGS_OK or a negative SGS_E* code; `sgs_last_error()` gives the
 *     thread-local message (the reference only ever raises RuntimeError/ValueError: the
 *     Python host layer turns a non-zero code into RuntimeError).
 */
#ifndef SGS_HIP_H_
#define SGS_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGS_ABI_VERSION 1

#define SGS_OK 0
#define SGS_EINVAL (-1)    /* bad argument (shape, null pointer, q > E ...) */
#define SGS_EWORKSPACE (-2) /* workspace too small */
#define SGS_EHIP (-3)      /* a HIP runtime call failed */

typedef void* sgs_stream_t;

int sgs_abi_version(void);
const char* sgs_last_error(void);

/* ------------------------------------------------------------------------------------
 * Counter-based randomness (replaces torch's global generator draws).
 *   sgs_exp_noise   : noise[e] ~ Exp(1), Philox4x32-10 keyed by `seed`, counter (e, stream_id).
 *                     The sampler generates exactly these values in-register when it is given
 *                     noise == NULL (reference: the exponential_() inside torch.multinomial,
 *                     sampling.py:96, training_hybrid.py:47).
 *   sgs_dropout_keep: keep[r*cols+c] in {0,1}, P(keep)=1-p, hash of (seed, site, r, c); the
 *                     same bits the fused kernels apply (reference: nn.Dropout at
 *                     model.py:107,121,160).  Only tests need the materialised form.
 * ---------------------------------------------------------------------------------- */
/* HIP-graph replay support: seeds are passed by value and therefore frozen in a captured graph.  When a
 * device word is registered here, every RNG-consuming kernel uses seed + 0x9E3779B97F4A7C15 * (*epoch_dev);
 * a captured increment of that word makes every replay draw fresh noise and dropout masks.  NULL (the
 * default) = seeds used exactly as given.  Process-wide; the word must outlive all launches that read it. */
int sgs_rng_set_epoch_buffer(const uint64_t* epoch_dev);
int sgs_exp_noise(uint64_t seed, uint64_t stream_id, int64_t E, float* noise, sgs_stream_t stream);
int sgs_dropout_keep(uint64_t seed, uint32_t site, int64_t rows, int64_t cols, float p, uint8_t* keep,
                     sgs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * One captured step for partitions of ANY size (HIP-graph replay of training_hybrid.py:29-187's loop body without one capture
 * per partition).  Sizes and pointers are launch arguments, frozen at capture; two entry points lift that:
 *   sgs_dyn_edges_set : while a device word is registered, the entry points whose grids run over the CANDIDATE edges of a partition
 *                       (sgs_sample_topq, sgs_edge_score_fwd, the dense part of sgs_st_weights_bwd, sgs_gather_by_eid /
 *                       sgs_scatter_by_eid) use n = min(their size argument, *word) rows and treat the argument as the CAPACITY
 *                       (grid size, row stride of `edge_index`, buffer and workspace sizes).  A drawn subgraph has q < E edges,
 *                       so calls over it are unaffected.  Process-wide, read at launch (= capture) time; NULL = off.
 *                       Everything downstream of the draw is sized by q and N, which a run fixes.
 *   sgs_stage_segments: the batch hand-over of training_hybrid.py:42 (`batch.to(device)`) for resident partitions: ONE launch copies
 *                       up to sgs_stage_max_segments() spans from a partition's resident arrays into the static buffers a
 *                       captured step reads, pads each destination tail with a 32-bit word (zero rows for padded nodes, E for
 *                       the row pointers past N, ...) and writes up to 4 dims words (the E that sgs_dyn_edges_set points at).
 *                       desc_host [n_segments][5] int64, HOST memory, read during the call: {src, dst, src_bytes, dst_bytes >=
 *                       src_bytes, pad word}; spans are whole 4-byte words.  HBM-bound: 16 B per lane, 16 KiB per workgroup.
 * ---------------------------------------------------------------------------------- */
int sgs_dyn_edges_set(const int64_t* n_edges_dev);
int sgs_stage_max_segments(void);
int sgs_stage_segments(const int64_t* desc_host, int64_t n_segments, int64_t* dims_dev, const int64_t* dims_host, int64_t n_dims,
                       sgs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K0 / K2 / K3: fused exponential-race top-q edge sampler + stable compaction.
 *
 * Replaces, in one call:
 *   mode SGS_SAMPLE_LEARNED  sampling.py:91-96,134-139 (`gumbel_softmax_sampling`: normalise,
 *                            mix with the degree prior unless istest, multinomial w/o
 *                            replacement, one-hot mask) + training_hybrid.py:83,86
 *                            (`edge_index[:, mask]`, `edge_probs_full[mask]`);
 *   mode SGS_SAMPLE_PRIOR    training_hybrid.py:46-48 (softmax(batch.prob), multinomial,
 *                            column gather) -- output is in ORIGINAL edge order (the
 *                            reference's race order is irrelevant to every consumer).
 *
 *   s_e   = p_e / (sum(p) + 1e-12)                      (learned)
 *         = (1-c) * s_e + c * prior_e                   (learned, prior != NULL, i.e. !istest)
 *         = exp(p_e - max p) / sum exp(p - max p)       (prior mode: softmax of `p`)
 *   key_e = s_e / noise_e        IEEE fp32 divide, noise ~ Exp(1)
 *   selected = the q largest keys; ties broken towards the LOWEST edge id.
 *
 * `noise` may be NULL: then noise_e is generated in-register exactly as sgs_exp_noise(seed,
 * stream_id) would.  Outputs (any may be NULL except mask): mask[E] (0/1 bytes, torch.bool
 * compatible), sampled_eid[q] ascending edge ids, sampled_edge_index[2,q], sampled_p[q] =
 * p[sampled_eid], stats[4] = {Z (sum p or sum exp), max (prior mode), threshold key, #ties
 * taken at the threshold}.  keys_out[E] (optional) receives the fp32 keys (tests only).
 * p == NULL (mode LEARNED, prior == NULL): uniform weights, i.e. a uniformly random q-subset of the edges -- the selection
 * behind `random_edge_sampling` (sampling.py:159-163, torch.randperm(E)[:q]), emitted in original edge order.
 *
 * sgs_gather_columns: out[:, j] = edge_index[:, idx[j]] (sampling.py:163 with an explicit index vector, e.g. the first q
 * entries of a given permutation); indices outside [0, E) give (-1, -1).
 * ---------------------------------------------------------------------------------- */
#define SGS_SAMPLE_LEARNED 0
#define SGS_SAMPLE_PRIOR 1

size_t sgs_sample_topq_workspace_bytes(int64_t E);
int sgs_sample_topq(int mode, const float* p, const float* prior, double degree_bias_coef,
                    const float* noise, uint64_t seed, uint64_t stream_id, int64_t E, int64_t q,
                    const int64_t* edge_index, uint8_t* mask, int64_t* sampled_eid,
                    int64_t* sampled_edge_index, float* sampled_p, float* stats, float* keys_out,
                    void* ws, size_t ws_bytes, sgs_stream_t stream);

int sgs_gather_columns(const int64_t* edge_index, int64_t E, const int64_t* idx, int64_t q, int64_t* out, sgs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Phase API of the sampler for EDGE-SHARDED draws (config 5: one graph, edges split over R ranks in
 * contiguous shards whose boundaries are multiples of sgs_sampler_chunk()).  Between phases the host
 * runs the collectives (torch.distributed over RCCL): all-gather of the per-chunk partial sums
 * (reduced by sgs_sampler_shard_finalize in the single-GPU order => bit-identical normaliser),
 * all-reduce(sum) of each 2048-bin digit histogram, all-gather of the per-rank (#greater, #equal)
 * counts.  Noise is keyed by the GLOBAL edge id (edge_offset + local id), ties go to the lowest global
 * ids, so the selected set is identical for every rank count (tests: 1 vs 2 vs 3 ranks).
 *   keys buffer: first region of the caller's workspace (sgs_sampler_shard_workspace_bytes), uint32 [E].
 *   state: 16 bytes of device memory, zeroed by the caller before pass 0.
 * ---------------------------------------------------------------------------------- */
size_t sgs_sampler_shard_workspace_bytes(int64_t E_local);
int64_t sgs_sampler_chunk(void);
int sgs_sampler_shard_partials(int stage, const float* p, int64_t E, const float* scal, float* part, sgs_stream_t stream);
int sgs_sampler_shard_finalize(int stage, const float* part_all, int64_t nblk_all, float* scal, sgs_stream_t stream);
int sgs_sampler_shard_keys(int mode, const float* p, const float* prior, double degree_bias_coef, const float* noise,
                           uint64_t seed, uint64_t stream_id, int64_t edge_offset, int64_t E, const float* scal,
                           uint32_t* keys, float* keys_out, uint32_t* hist, sgs_stream_t stream);
int sgs_sampler_shard_hist(const uint32_t* keys, int64_t E, int pass, const void* state, uint32_t* hist,
                           sgs_stream_t stream);
int sgs_sampler_select(uint32_t* hist, int pass, int64_t q, void* state, sgs_stream_t stream);
int sgs_sampler_shard_count(const uint32_t* keys, int64_t E, const void* state, uint32_t* counts, void* ws,
                            size_t ws_bytes, sgs_stream_t stream);
int sgs_sampler_shard_compact(const uint32_t* keys, int64_t E, const void* state, int64_t ties_local, int64_t q_local,
                              int64_t edge_offset, const float* p, const int64_t* edge_index_local, uint8_t* mask,
                              int64_t* sampled_eid, int64_t* sampled_edge_index, float* sampled_p, void* ws,
                              size_t ws_bytes, sgs_stream_t stream);

/* Straight-through weights of sampling.py:137-138,155 for the selected edges:
 *   w_j = clamp(p_e * ((1 - s_e) + s_e), 0, 1),  e = sampled_eid[j]   (forward value)
 * and its exact autograd backward wrt p (s depends on every p through sum(p)):
 *   dp_e += [e selected] g_e ((1 - s_e) + s_e) + ... (see DESIGN.md "straight-through").
 * Used by --pipeline straight_through and by evaluate.py.  prior == NULL <=> istest. */
int sgs_st_weights_fwd(const float* p, const float* prior, double degree_bias_coef, const float* stats,
                       const int64_t* sampled_eid, int64_t E, int64_t q, float* w, sgs_stream_t stream);
size_t sgs_st_weights_bwd_workspace_bytes(int64_t E, int64_t q);
int sgs_st_weights_bwd(const float* p, const float* prior, double degree_bias_coef, const float* stats,
                       const int64_t* sampled_eid, const float* grad_w, int64_t E, int64_t q, float* grad_p,
                       void* ws, size_t ws_bytes, sgs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K4: graph preparation for the GCN layers = the bookkeeping half of PyG's
 * add_remaining_self_loops + gcn_norm, done ONCE per sampled graph and shared by both layers,
 * forward and backward (the reference redoes it inside every GCNConv call: model.py:107-111,
 * 159-161).
 *
 * For the n_edges edges (src,dst) of `edge_index` [2,n_edges] over N nodes, builds both CSR
 * orientations, rows ordered by edge id (deterministic, so every floating-point row sum is
 * run-to-run reproducible):
 *   in_ptr[N+1],  in_src[n_edges],  in_eid[n_edges]    rows = dst   (forward aggregation)
 *   out_ptr[N+1], out_dst[n_edges], out_eid[n_edges]   rows = src   (transposed / backward)
 *   loop_eid[N] = id of the LAST existing self-loop edge of node i, or -1  (PyG: an existing
 *                 loop keeps its weight, otherwise the added loop has weight 1)
 * Self-loop edges stay in the CSRs (the edge scorer's backward needs them); the norm kernel
 * gives them weight 0 there and routes them through the loop term.  int32 indices.
 * ---------------------------------------------------------------------------------- */
size_t sgs_graph_build_workspace_bytes(int64_t n_edges, int64_t N);
int sgs_graph_build(const int64_t* edge_index, int64_t n_edges, int64_t N, int32_t* in_ptr, int32_t* in_src,
                    int32_t* in_eid, int32_t* out_ptr, int32_t* out_dst, int32_t* out_eid, int32_t* loop_eid,
                    void* ws, size_t ws_bytes, sgs_stream_t stream);

/* CSR of a sampled subgraph from its PARENT's CSR (sgs_graph_build of the partition, built once and cached): the draw keeps
 * a subset of the parent's edges in their original order, so the child rows are the parent rows with the unselected entries
 * squeezed out and edge ids renumbered by rank in `sampled_eid` (ascending parent edge ids, as sgs_sample_topq emits them):
 * no atomics, no per-row sort, three launches.  mask [E_parent] u8; child arrays sized as for sgs_graph_build(q, N).
 * Result is identical to sgs_graph_build on the compacted edge list. */
size_t sgs_graph_filter_workspace_bytes(int64_t E_parent, int64_t N);
int sgs_graph_filter(const int32_t* pin_ptr, const int32_t* pin_src, const int32_t* pin_eid, const int32_t* pout_ptr,
                     const int32_t* pout_dst, const int32_t* pout_eid, int64_t E_parent, int64_t N, const uint8_t* mask,
                     const int64_t* sampled_eid, int64_t q, int32_t* in_ptr, int32_t* in_src, int32_t* in_eid, int32_t* out_ptr,
                     int32_t* out_dst, int32_t* out_eid, int32_t* loop_eid, void* ws, size_t ws_bytes, sgs_stream_t stream);

/* sgs_graph_build for an edge list that is SORTED BY SOURCE (what a draw over a row-sorted edge list emits; the reference's loaders all
 * produce row-sorted lists, and sampling.py keeps the order: edge_index[:, mask]).  The out-CSR is then the list itself; the in-CSR is ONE
 * stable radix sort of (dst, (src, edge id)) -- at whole-graph scale (config 5: q = 22.9 M drawn edges of 114.6 M) this replaces
 * sgs_graph_filter's passes over the PARENT's CSR.  Same arrays as sgs_graph_build.  `unsorted` (device word, may be NULL): set to 1 if
 * the list turns out not to be sorted by source (the arrays are then meaningless), 0 otherwise. */
size_t sgs_graph_build_src_sorted_workspace_bytes(int64_t n_edges, int64_t N);
int sgs_graph_build_src_sorted(const int64_t* edge_index, int64_t n_edges, int64_t N, int32_t* in_ptr, int32_t* in_src, int32_t* in_eid,
                               int32_t* out_ptr, int32_t* out_dst, int32_t* out_eid, int32_t* loop_eid, int32_t* unsorted, void* ws,
                               size_t ws_bytes, sgs_stream_t stream);

/* gcn_norm forward (PyG gcn_norm, add_self_loops=True, flow source_to_target):
 *   loopw_i = w[loop_eid_i] or 1;  deg_i = loopw_i + sum_{e=(j->i), j!=i} w_e;  dis = deg^-1/2 (inf -> 0)
 *   what_in[k]  = dis[src] * w * dis[dst] in dst-CSR order,  what_out[k] the same in src-CSR order
 *   what_loop[i] = dis_i * loopw_i * dis_i
 * w == NULL means unit weights. */
int sgs_gcn_norm_fwd(const float* w, int64_t n_edges, int64_t N, const int32_t* in_ptr, const int32_t* in_src,
                     const int32_t* in_eid, const int32_t* out_ptr, const int32_t* out_dst, const int32_t* out_eid,
                     const int32_t* loop_eid, float* dis, float* loopw, float* what_in, float* what_out,
                     float* what_loop, sgs_stream_t stream);

/* Edge-sharded gcn_norm (config 5): each rank sums the weights of its own in-edges
 * (sgs_gcn_degree_partial), the host all-reduces the [N] vector, and sgs_gcn_norm_from_degree continues
 * with deg_i = 1 + degsum_i (the added self loop counts once; graphs with existing (i,i) edges are not
 * supported in sharded mode).  sgs_bias_act applies the layer epilogue AFTER the all-reduce of the
 * partial aggregates:  Y = act(X + bias), act as in sgs_spmm_csr. */
int sgs_gcn_degree_partial(const float* w, int64_t n_edges, int64_t N, const int32_t* in_ptr, const int32_t* in_src,
                           const int32_t* in_eid, float* degpart, sgs_stream_t stream);
int sgs_gcn_norm_from_degree(const float* w, const float* degsum, int64_t n_edges, int64_t N, const int32_t* in_ptr,
                             const int32_t* in_src, const int32_t* in_eid, const int32_t* out_ptr,
                             const int32_t* out_dst, const int32_t* out_eid, float* dis, float* loopw, float* what_in,
                             float* what_out, float* what_loop, sgs_stream_t stream);
int sgs_bias_act(const float* X, const float* bias, int64_t N, int64_t D, int act, float p_drop, uint64_t seed,
                 uint32_t site, float* Y, sgs_stream_t stream);
/* The same on a BLOCK of rows of a larger matrix (node-block sharding): X, Y [n_rows, D] hold rows row_offset .. row_offset + n_rows - 1,
 * and the dropout rows are those global ids, so the mask equals the unsharded one's. */
int sgs_bias_act_rows(const float* X, const float* bias, int64_t n_rows, int64_t D, int64_t row_offset, int act, float p_drop,
                      uint64_t seed, uint32_t site, float* Y, sgs_stream_t stream);

/* GraphSAGE mean aggregation for the GSAGE scorer (model.py:47-89, PyG SAGEConv aggr='mean'): per-entry
 * weights 1 / indeg(dst) in both CSR orders, to be used with sgs_spmm_csr (diag = NULL).
 * sgs_degree_prior_logits: the argument of the softmax in datasets.py:141-156 (`add_degree`),
 * E^-1/2 / (colcount[row_e] + rowcount[col_e] + 1e-10); the softmax itself is the sampler's prior mode,
 * or torch.softmax when the materialised `data.prob` is wanted. */
int sgs_mean_weights(int64_t n_edges, int64_t N, const int32_t* in_ptr, const int32_t* out_ptr, const int32_t* out_dst,
                     float* what_in, float* what_out, sgs_stream_t stream);
int sgs_degree_prior_logits(const int64_t* edge_index, int64_t E, int64_t N, const int32_t* in_ptr,
                            const int32_t* out_ptr, float* logits, sgs_stream_t stream);

/* gcn_norm backward: from gw_hat[e] = dL/d(what_e) (edge-id order) and gloop[i] = dL/d(what_loop_i)
 * to dL/dw_e, through both the message weight and the degree normalisation:
 *   dw_e = gw_hat_e dis_s dis_t + Hn_t,            Hn_t = -1/2 dis_t^3 G_t
 *   G_t  = sum_{e' into t} gw_hat_e' w_e' dis_src + sum_{e' out of t} gw_hat_e' w_e' dis_dst
 *          + 2 gloop_t loopw_t dis_t
 * (every existing self-loop edge (i,i) gets gloop_i dis_i^2 + Hn_i -- also a duplicate whose weight was
 * overwritten, which is what autograd does for PyG's `loop_attr[idx] = edge_attr[inv_mask]`). */
size_t sgs_gcn_norm_bwd_workspace_bytes(int64_t N);
int sgs_gcn_norm_bwd(const float* w, const float* gw_hat, const float* gloop, int64_t n_edges, int64_t N,
                     const float* dis, const float* loopw, const int32_t* in_ptr, const int32_t* in_src,
                     const int32_t* in_eid, const int32_t* out_ptr, const int32_t* out_dst, const int32_t* out_eid,
                     const int32_t* loop_eid, const int64_t* edge_index, float* dw, void* ws, size_t ws_bytes,
                     sgs_stream_t stream);
/* ... with up to two upstream gradients summed on read (gw_hat2 / gloop2: the second GCN layer over the same normalisation; both or neither)
 * and another consumer's d w added on the way out (dw_add, optional; may be dw itself) -- the add kernels autograd would launch. */
int sgs_gcn_norm_bwd_sum(const float* w, const float* gw_hat, const float* gloop, const float* gw_hat2, const float* gloop2, const float* dw_add,
                         int64_t n_edges, int64_t N, const float* dis, const float* loopw, const int32_t* in_ptr, const int32_t* in_src,
                         const int32_t* in_eid, const int32_t* out_ptr, const int32_t* out_dst, const int32_t* out_eid,
                         const int32_t* loop_eid, const int64_t* edge_index, float* dw, void* ws, size_t ws_bytes, sgs_stream_t stream);

/* Split form of sgs_gcn_norm_bwd for edge-sharded graphs: Hn (per node) is linear in the edge contributions, so
 * each rank computes its partial Hn (gloop non-zero only on the rank that owns the self-loop term), the host
 * all-reduces Hn [N], and the per-edge pass finishes locally. */
int sgs_gcn_norm_bwd_node(const float* w, const float* gw_hat, const float* gloop, int64_t n_edges, int64_t N,
                          const float* dis, const float* loopw, const int32_t* in_ptr, const int32_t* in_src,
                          const int32_t* in_eid, const int32_t* out_ptr, const int32_t* out_dst, const int32_t* out_eid,
                          float* Hn, sgs_stream_t stream);
int sgs_gcn_norm_bwd_edge(const float* gw_hat, const float* gloop, int64_t n_edges, int64_t N, const float* dis,
                          const int32_t* loop_eid, const int64_t* edge_index, const float* Hn, float* dw,
                          sgs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K5: weighted CSR SpMM with fused epilogue (GCNConv propagate + bias, F.relu, nn.Dropout;
 * model.py:107-111,159-161):
 *   Y[i,:] = act( sum_{k in row i} val[k] X[col[k],:] + diag[i] X[i,:] + bias )
 * act: identity | ReLU | ReLU then counter-based dropout(p, seed, site) (see sgs_dropout_keep).
 * Forward uses (in_ptr, in_src, what_in, what_loop); the transposed product of backward,
 * dX = A_hat^T dZ, is the same call with (out_ptr, out_dst, what_out, what_loop).
 * diag / bias may be NULL; nnz = ptr[N] (known to the caller; picks the row-per-workgroup kernel for
 * few long rows).  X is the already linearly transformed feature matrix [N,D]
 * (the dense X W^T is a library GEMM on the host side).
 * ---------------------------------------------------------------------------------- */
#define SGS_ACT_NONE 0
#define SGS_ACT_RELU 1
#define SGS_ACT_RELU_DROPOUT 2
int sgs_spmm_csr(const float* X, int64_t N, int64_t D, int64_t nnz, const int32_t* ptr, const int32_t* col, const float* val,
                 const float* diag, const float* bias, int act, float p_drop, uint64_t seed, uint32_t site, float* Y,
                 sgs_stream_t stream);

/* SDDMM over the same CSR (gradient wrt the normalised weights):
 *   g[eid[k]] = <A[i,:], B[col[k],:]> for k in row i;  gdiag[i] = <A[i,:], B[i,:]>  (gdiag may be NULL) */
int sgs_sddmm_csr(const float* A, const float* B, int64_t N, int64_t D, int64_t nnz, const int32_t* ptr, const int32_t* col,
                  const int32_t* eid, float* g, float* gdiag, sgs_stream_t stream);

/* dZ = dY * act'(Y) for the fused epilogue above (Y is the layer OUTPUT: Y > 0 iff kept and
 * positive, so no mask is stored);  colsum: out[d] = sum_i A[i,d]  (bias gradient). */
int sgs_act_bwd(const float* dY, const float* Y, int64_t n, int act, float p_drop, float* dZ, sgs_stream_t stream);
size_t sgs_colsum_workspace_bytes(int64_t N, int64_t D);
int sgs_colsum(const float* A, int64_t N, int64_t D, float* out, void* ws, size_t ws_bytes, sgs_stream_t stream);
/* Both in one pass (one launch for partition-sized N): dZ = dY * act'(Y) written, colsum[d] = sum_i dZ[i,d] -- the activation and bias
 * gradients of one layer's backward (autograd of model.py:159-161).  ws: sgs_colsum_workspace_bytes(N, D). */
int sgs_act_bwd_colsum(const float* dY, const float* Y, int64_t N, int64_t D, int act, float p_drop, float* dZ, float* colsum, void* ws,
                       size_t ws_bytes, sgs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K1b: fused edge scorer (model.py:29-34 / 115-122 `_edge_score`; never materialises the
 * reference's [E,2H] feature or [E,H] hidden tensors):
 *   p_e = sigmoid( w2 . drop(relu( W1 [x_s*x_d | x_s-x_d] + b1 )) + b2 ),  x = codes rows
 * Algebraic split  W1 [x*y | x-y] = W1a (x*y) + U[s] - U[d]  with U = codes W1b^T a node-level
 * library GEMM done by the caller, so the per-edge contraction is H x H; it runs on the f32
 * matrix cores (v_mfma_f32_32x32x2_f32, exact fp32).
 *   codes, U [N,H] f32; W1 = fc1.weight [H,2H]; b1 [H]; w2 = fc2.weight [H]; b2 [1].
 *   4 <= H <= 256, H % 4 == 0.  Dropout on the hidden layer is counter-based, row = edge_id_offset + local
 *   edge id (edge_id_offset = 0 unless the edge list is a shard of a larger graph).
 * ws: sgs_edge_score_workspace_bytes(N, H, E).
 *
 * sgs_edge_score_bwd_core runs over an explicit list of active edges (hybrid / two-pass: the q
 * sampled edges -- every other edge has exactly zero upstream gradient; NULL = all E edges in
 * order) and RECOMPUTES the hidden layer instead of storing it.  It writes, per active row j:
 *   dv[j,:]  = dL/d(fc1 pre-activation)       dz[j] = grad_p_j p_j (1-p_j)  (sum -> d b2)       feat[j,:] = x_s * x_d
 * and, per tile of sgs_edge_score_bwd_tile() (= 64) consecutive active rows, ONE row of
 *   hdz_part[tile,:] = sum_{j in tile} dz_j * hidden_j          (rows sum to d w2; hidden is never written per edge)
 * from which the host forms  d W1a = dv^T feat (sgs_gemm_tn),  d b1 = colsum(dv),  d w2 = colsum(hdz_part),
 * dfeat = dv W1a (library GEMM) and the two endpoint reductions below.
 *
 * sgs_endpoint_reduce: out[v,:] = sum_{k in out-row v} sign_out M_out[out_eid[k],:] (* T[out_dst[k],:])
 *                               + sum_{k in in-row v}  sign_in  M_in[in_eid[k],:]   (* T[in_src[k],:])
 * over the CSRs of the ACTIVE edge list (sgs_graph_build): the scatter of per-edge gradient rows
 * to both endpoints as a deterministic gather (no float atomics).  T may be NULL.
 *   d codes (direct) = reduce(dfeat, dfeat, T = codes, +1, +1);   d U = reduce(dv, dv, NULL, +1, -1).
 * ---------------------------------------------------------------------------------- */
size_t sgs_edge_score_workspace_bytes(int64_t N, int64_t H, int64_t E);   /* E = 0 for the backward core */
int sgs_edge_score_get_variant(void);          /* the overrides currently set (-1 = automatic) */
int sgs_edge_score_get_bwd_variant(void);
void sgs_edge_score_set_bwd_variant(int variant); /* backward core: -1 = automatic (4 at H % 128 == 0 and >= 65 536 active rows, else 0), 0 = LDS-tiled, 3 = 64-edge streaming loop (A/B: measured slower), 4 = bf16x6 loop */
int sgs_edge_score_bwd_tile(void);              /* active rows per hdz_part row (64) */
void sgs_edge_score_set_variant(int variant);   /* forward kernel: -1 = automatic (default: when E >= 65 536, 4 if H % 128 == 0 else 3; below that 1),
                                                  * 0 = LDS-tiled, 1 = register-streaming (32-edge wave tile), 2 = weight-stationary
                                                  * persistent, 3 = register-streaming with a 64-edge wave tile, 4 = bf16x6: exact
                                                  * 3-way bf16 splits of both operands, six v_mfma_f32_32x32x16_bf16 per fp32
                                                  * product (fp32-faithful; H % 128 == 0, else 3); all give the same p to fp32
                                                  * rounding */
int sgs_edge_score_fwd(const float* codes, const float* U, int64_t N, int64_t H, const int64_t* edge_index, int64_t E,
                       int64_t edge_id_offset, const float* W1, const float* b1, const float* w2, const float* b2,
                       float p_drop, uint64_t seed, uint32_t site, float* p_out, void* ws, size_t ws_bytes,
                       sgs_stream_t stream);
/* Paired forward.  fc1's edge-level half W1a (x_s * x_d) is symmetric in the endpoints, so on an undirected graph stored in both
 * directions (every dataset of the reference: datasets.py:189-190) an edge and its reverse share that contraction bit for bit; they
 * differ in the sign of U[s] - U[d] and in their dropout rows.  sgs_edge_mates pairs the edges (mate[e] = id of (dst_e -> src_e)
 * or -1; mutual, one to one; needs the src-CSR of sgs_graph_build), the caller lists the canonical edges (mate < 0 or e < mate)
 * and sgs_edge_score_fwd_paired runs the H x H contraction for those M edges only, finishing both scores of a pair in its
 * epilogue: p_out [E] equals sgs_edge_score_fwd's bit for bit.  H = 128 or 256 (ask sgs_edge_score_paired_supported).
 * Under sgs_dyn_edges_set the live M is read from word 1 of the registered dims (word 0 = live E). */
size_t sgs_edge_mates_workspace_bytes(int64_t n_edges);
int sgs_edge_mates(const int64_t* edge_index, int64_t n_edges, int64_t N, const int32_t* out_ptr, const int32_t* out_dst,
                   const int32_t* out_eid, int32_t* mate, void* ws, size_t ws_bytes, sgs_stream_t stream);
int sgs_edge_score_paired_supported(int64_t H);
int sgs_edge_score_fwd_paired(const float* codes, const float* U, int64_t N, int64_t H, const int64_t* edge_index, int64_t E,
                              int64_t edge_id_offset, const int32_t* canon, int64_t M, const int32_t* mate, const float* W1, const float* b1,
                              const float* w2, const float* b2, float p_drop, uint64_t seed, uint32_t site, float* p_out, void* ws,
                              size_t ws_bytes, sgs_stream_t stream);
int sgs_edge_score_bwd_core(const float* codes, const float* U, int64_t N, int64_t H, const int64_t* edge_index,
                            int64_t E, int64_t edge_id_offset, const int64_t* active_eid, int64_t n_active, const float* grad_p,
                            const float* W1, const float* b1, const float* w2, const float* b2, float p_drop,
                            uint64_t seed, uint32_t site, float* dv, float* hdz_part, float* dz, float* feat, void* ws,
                            size_t ws_bytes, sgs_stream_t stream);
/* dfeat [n,H] = dv [n,H] . W1[:, :H]: the gradient wrt the Hadamard features x_s * x_d that the scorer backward (model.py:29-34 under
 * autograd) forms from sgs_edge_score_bwd_core's dv.  The forward's bf16x6 loop as a row GEMM (fp32-faithful, see
 * sgs_edge_score_set_variant); H = 128 or 256 only -- ask sgs_edge_score_bwd_dfeat_supported, other sizes use a library GEMM.
 * Workspace: sgs_edge_score_workspace_bytes(0, H, 0). */
int sgs_edge_score_bwd_dfeat_supported(int64_t H);
int sgs_edge_score_bwd_dfeat(const float* dv, int64_t n, int64_t H, const float* W1, float* dfeat, void* ws, size_t ws_bytes,
                             sgs_stream_t stream);

/* The MASK form of the scorer backward (H = 128 or 256; sgs_edge_score_bwd_bits_supported).  With v the fc1 pre-activation,
 *   dv[r, h] = dz[r] * w2[h] * [dropout(relu(v))[r, h] > 0] / (1 - p)            (autograd of model.py:31-33 / 119-121)
 * is a 0 / 1 matrix times a row and a column factor, so the core writes ONE BIT per entry -- dvbits [n, H/32], bit h of row r in word
 * h / 32 -- instead of the fp32 [n, H] matrix, and the three consumers take bits + dz + w2:
 *   sgs_edge_score_bwd_dfeat_bits   dfeat = dv W1a            = dz[r] * (bits[r, :] . diag(w2 / (1 - p)) W1a)
 *   sgs_gemm_tn_mask                d W1a = dv^T feat, d b1   = diag(w2 / (1 - p)) (bits^T (diag(dz) feat))
 *   sgs_endpoint_reduce_pair_bits   d U[v] = w2 / (1 - p) * (sum_out - sum_in) dz[e] bits[e, :]   (+ the d codes half, as sgs_endpoint_reduce_pair)
 * A 0 / 1 operand is exact in bf16: the two contractions issue 3 bf16 MFMA products per fp32 product (the other operand's exact 3-way
 * split) instead of 6, still fp32-faithful.  Other arguments as sgs_edge_score_bwd_core / _bwd_dfeat / sgs_endpoint_reduce_pair. */
int sgs_edge_score_bwd_bits_supported(int64_t H);
/* No recompute at all when the FORWARD kept the mask (sgs_edge_score_fwd_mask: the bf16x6 forward, paired when canon / mate are given, that
 * also writes maskbits [E, H/32] for every scored edge; scores bit-identical to the plain forward):
 *   sgs_edge_score_bwd_prep        per active row r (edge e): dz[r] = grad_p[r] p[e] (1 - p[e]), dvbits[r, :] = maskbits[e, :],
 *                                  feat[r, :] = codes[src e, :] * codes[dst e, :]                       (one HBM-bound pass)
 *   sgs_edge_score_dw2_from_parts  d fc2.weight[h] = 1/(1-p) * ( sum_k W1a[h,k] T[h,k] + sum_v U[v,h] R[v,h] + b1[h] c[h] ) from the
 *                                  consumers' results before their factors: T = C_raw and c = colsum_raw of sgs_gemm_tn_mask,
 *                                  R = out_U_raw of sgs_endpoint_reduce_pair_bits (exact algebra: hidden = mask * (W1a feat + U[s] - U[d] + b1) / (1-p))
 * then sgs_edge_score_bwd_dfeat_bits, sgs_gemm_tn_mask, sgs_endpoint_reduce_pair_bits as after sgs_edge_score_bwd_core_bits. */
int sgs_edge_score_fwd_mask(const float* codes, const float* U, int64_t N, int64_t H, const int64_t* edge_index, int64_t E,
                            int64_t edge_id_offset, const int32_t* canon, int64_t M, const int32_t* mate, const float* W1, const float* b1,
                            const float* w2, const float* b2, float p_drop, uint64_t seed, uint32_t site, float* p_out, uint32_t* maskbits,
                            void* ws, size_t ws_bytes, sgs_stream_t stream);
int sgs_edge_score_bwd_prep(const float* codes, int64_t N, int64_t H, const int64_t* edge_index, int64_t E, const int64_t* active_eid,
                            int64_t n_active, const float* grad_p, const float* p, const uint32_t* maskbits, float* dz, uint32_t* dvbits,
                            float* feat, sgs_stream_t stream);
int sgs_edge_score_dw2_from_parts(const float* W1, const float* T_raw, const float* U, const float* R_raw, const float* b1, const float* c_raw,
                                  int64_t N, int64_t H, float p_drop, float* dw2, sgs_stream_t stream);
/* ---- Endpoint-dropout scorer: EdgeProbMLP with dropout > 0 (model.py:16-45) ----
 * The reference drops the two endpoint codes of every scored edge independently, x = drop(relu(fcdim(X[src]))), y = drop(relu(fcdim(X[dst]))),
 * before `_edge_score(x, y)`.  relu(fcdim(.)) is row-wise and hoisted to the nodes (A [N, H]); the masks are per (edge, endpoint):
 *   keep_x[e, c] = sgs_dropout_keep(seed_x, site_x, row e, col c),  keep_y likewise,  x_m = A[src e] * keep_x / (1 - p_ep),  y_m = ... .
 * Masked codes differ per edge, so fc1's node-level half U[s] - U[d] does not exist: the kernels contract all 2H features [x_m * y_m | x_m - y_m]
 * against W1 [H, 2H] per edge (fp32 MFMA, LDS-tiled) and never materialise an [E, H] array in the forward.  H % 16 == 0.
 *   sgs_edge_score_epd_fwd       p_out [E]
 *   sgs_edge_score_epd_bwd_core  over the active rows: dv [n, H], hdz_part [cdiv(n, 64), H], dz [n], feat2 [n, 2H] (the features)
 *                                (then d W1 = dv^T feat2 by sgs_gemm_tn, dfeat2 = dv W1 by a library GEMM)
 *   sgs_edge_score_epd_reduce    d A [N, H] from dfeat2 [n, 2H] over both CSR orientations of the active edge list (masks recomputed)
 * ws: sgs_edge_score_epd_workspace_bytes(H). */
/* Probe knobs of the bf16x6 scorer kernels (tools/stagger_probe.py, tools/stagger_trace.py; the product path never calls them).
 * `stagger` >= 0 forces the start-up sleep of every CU's second resident workgroup (in 64-cycle quanta; -1 = the built-in policy),
 * `prio` = 1 runs the main loop at raised wave priority, 2 the epilogue; `mode_mask` = bit m set: kernel MODE m staggers. */
int sgs_edge_score_probe_set(int stagger, int prio, uint32_t mode_mask);
/* `buf` (device, 8 words per workgroup of the next launches, or NULL = off): shader-clock stamps at kernel start, main loop,
 * epilogue and end, and the hardware id of the CU the workgroup ran on. */
int sgs_edge_score_probe_trace(unsigned long long* buf);
size_t sgs_edge_score_epd_workspace_bytes(int64_t H);
int sgs_edge_score_epd_fwd(const float* A, int64_t N, int64_t H, const int64_t* edge_index, int64_t E, int64_t edge_id_offset, const float* W1,
                           const float* b1, const float* w2, const float* b2, float p_hidden, uint64_t seed, uint32_t site, float p_ep,
                           uint64_t seed_x, uint32_t site_x, uint64_t seed_y, uint32_t site_y, float* p_out, void* ws, size_t ws_bytes,
                           sgs_stream_t stream);
int sgs_edge_score_epd_bwd_core(const float* A, int64_t N, int64_t H, const int64_t* edge_index, int64_t E, int64_t edge_id_offset,
                                const int64_t* active_eid, int64_t n_active, const float* grad_p, const float* W1, const float* b1, const float* w2,
                                const float* b2, float p_hidden, uint64_t seed, uint32_t site, float p_ep, uint64_t seed_x, uint32_t site_x,
                                uint64_t seed_y, uint32_t site_y, float* dv, float* hdz_part, float* dz, float* feat2, void* ws, size_t ws_bytes,
                                sgs_stream_t stream);
int sgs_edge_score_epd_reduce(const float* dfeat2, const float* A, int64_t N, int64_t H, const int32_t* in_ptr, const int32_t* in_src,
                              const int32_t* in_eid, const int32_t* out_ptr, const int32_t* out_dst, const int32_t* out_eid,
                              const int64_t* active_eid, int64_t edge_id_offset, float p_ep, uint64_t seed_x, uint32_t site_x, uint64_t seed_y,
                              uint32_t site_y, float* dA, sgs_stream_t stream);

/* FUSED form of the same backward (round 3), for active rows SORTED BY SOURCE (a drawn subset of a row-sorted edge list, in edge order: every
 * dataset of the reference, datasets.py:189-190 to_undirected emits a coalesced list).  Neither feat nor dfeat exists as an [n, H] array:
 *   sgs_edge_score_bwd_prep_sd        as sgs_edge_score_bwd_prep without feat; sd [n, 2] int32 = the endpoints of every active row
 *   sgs_edge_score_bwd_dfeat_fused    the dfeat contraction with the by-source half of d codes reduced in its epilogue:
 *                                       G [n, H] = dfeat * codes[src]  (for the by-destination half),
 *                                       opart [sgs_edge_score_bwd_fused_opart_rows(n, N), H]: row (r >> 5) + src(r) = the sum of dfeat * codes[dst]
 *                                       over the rows of src(r) inside 32-row tile r >> 5, written by the LAST such row (other rows of opart: undefined)
 *   sgs_gemm_tn_mask_gather           d W1a with feat = codes[src] * codes[dst] gathered per row (gemm_tn section below)
 *   sgs_edge_score_bwd_reduce_fused   d codes[v] = sum of v's opart rows + sum_{in-row v} G;  d U / out_U_raw as sgs_endpoint_reduce_pair_bits
 * Results equal the unfused entry points' up to fp32 summation order.  HBM traffic at n = 100 000, H = 256: ~0.5 GB -> ~0.22 GB. */
int sgs_edge_score_bwd_prep_sd(const float* codes, int64_t N, int64_t H, const int64_t* edge_index, int64_t E, const int64_t* active_eid,
                               int64_t n_active, const float* grad_p, const float* p, const uint32_t* maskbits, float* dz, uint32_t* dvbits,
                               int32_t* sd, sgs_stream_t stream);
size_t sgs_edge_score_bwd_fused_opart_rows(int64_t n, int64_t N);
int sgs_edge_score_bwd_dfeat_fused(const uint32_t* dvbits, const float* dz, const int32_t* sd, const float* codes, int64_t n, int64_t N, int64_t H,
                                   const float* W1, const float* w2, float p_drop, float* G, float* opart, void* ws, size_t ws_bytes,
                                   sgs_stream_t stream);
int sgs_edge_score_bwd_reduce_fused(const float* G, const float* opart, const uint32_t* dvbits, const float* dz, const float* w2, float p_drop,
                                    int64_t N, int64_t H, int64_t nnz, const int32_t* in_ptr, const int32_t* in_eid, const int32_t* out_ptr,
                                    float* out_codes, float* out_U, float* out_U_raw, sgs_stream_t stream);
int sgs_edge_score_bwd_core_bits(const float* codes, const float* U, int64_t N, int64_t H, const int64_t* edge_index, int64_t E,
                                 int64_t edge_id_offset, const int64_t* active_eid, int64_t n_active, const float* grad_p, const float* W1,
                                 const float* b1, const float* w2, const float* b2, float p_drop, uint64_t seed, uint32_t site,
                                 uint32_t* dvbits, float* hdz_part, float* dz, float* feat, void* ws, size_t ws_bytes, sgs_stream_t stream);
int sgs_edge_score_bwd_dfeat_bits(const uint32_t* dvbits, const float* dz, int64_t n, int64_t H, const float* W1, const float* w2, float p_drop,
                                  float* dfeat, void* ws, size_t ws_bytes, sgs_stream_t stream);
int sgs_endpoint_reduce_pair_bits(const float* dfeat, const uint32_t* dvbits, const float* dz, const float* w2, float p_drop, const float* codes,
                                  int64_t N, int64_t H, int64_t nnz, const int32_t* in_ptr, const int32_t* in_src, const int32_t* in_eid,
                                  const int32_t* out_ptr, const int32_t* out_dst, const int32_t* out_eid, float* out_codes, float* out_U,
                                  float* out_U_raw, sgs_stream_t stream);
/* C[M, N] (row stride ldc) = (diag(dz) mask diag(rowscale * scale))^T B, mask bits [K, M/32]; colsum_A (optional, [M]) = that matrix's column
 * sums; dz_sum (optional, [1]) = sum_k dz[k] (d fc2.bias rides along); C_raw [M, N] / colsum_raw [M] (optional): the same two results
 * before the factor rowscale * scale (terms of d fc2.weight, sgs_edge_score_dw2_from_parts).  Tall-K shapes only (sgs_gemm_tn_mask_supported);
 * ws: sgs_gemm_tn_workspace_bytes(K, M, N). */
int sgs_gemm_tn_mask_supported(int64_t K, int64_t M, int64_t N);
/* ... with B never materialised: row k of B = codes[src k, :] * codes[dst k, :], (src, dst) = sd[k] (int32 [K, 2]), `codes` [codes_rows, N] row-major (codes_rows * N < 2^32).
 * Bit-identical to sgs_gemm_tn_mask on the materialised rows (same products, same order). */
int sgs_gemm_tn_mask_gather(const uint32_t* Abits, const float* dz, const float* rowscale, float scale, const float* codes, int64_t codes_rows,
                            const int32_t* sd, int64_t K, int64_t M, int64_t N, float* C, int64_t ldc, float* colsum_A, float* dz_sum, float* C_raw,
                            float* colsum_raw, void* ws, size_t ws_bytes, sgs_stream_t stream);
/* A/B switch of sgs_gemm_tn_mask_gather (tests, tools): shared = 1 (default) the shared-operand kernel -- the four waves of a K-group gather,
 * multiply and split a step's rows ONCE and exchange the operand fragments through LDS --, 0 the per-wave-slice kernel of round 2 (also what
 * shapes the shared kernel does not serve fall back to).  slabs > 0 forces the number of K-slices (workgroups per column group). */
void sgs_gemm_tn_set_gather_variant(int shared, int slabs);
int sgs_gemm_tn_mask(const uint32_t* Abits, const float* dz, const float* rowscale, float scale, const float* B, int64_t K, int64_t M, int64_t N,
                     float* C, int64_t ldc, float* colsum_A, float* dz_sum, float* C_raw, float* colsum_raw, void* ws, size_t ws_bytes,
                     sgs_stream_t stream);

int sgs_endpoint_reduce(const float* M_out, const float* M_in, const float* T, int64_t N, int64_t H, int64_t nnz,
                        const int32_t* in_ptr, const int32_t* in_src, const int32_t* in_eid, const int32_t* out_ptr,
                        const int32_t* out_dst, const int32_t* out_eid, float sign_out, float sign_in, float* out,
                        sgs_stream_t stream);
/* Both endpoint reductions of the scorer backward in one pass (needs H % 4 == 0, N <= 65536; 16-byte aligned rows):
 *   out_codes = reduce(dfeat, dfeat, T = codes, +1, +1)        out_U = reduce(dv, dv, NULL, +1, -1) */
int sgs_endpoint_reduce_pair(const float* dfeat, const float* dv, const float* codes, int64_t N, int64_t H, int64_t nnz,
                             const int32_t* in_ptr, const int32_t* in_src, const int32_t* in_eid, const int32_t* out_ptr,
                             const int32_t* out_dst, const int32_t* out_eid, float* out_codes, float* out_U, sgs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K6: gate and losses (training_hybrid.py:92-133, utils.py:163-169, 187-211), all on device.
 * ---------------------------------------------------------------------------------- */
/* correct[0] = #{i in train : argmax_c logits[i,c] == y_i} (first maximum wins), correct[1] = #train.
 * micro-F1 of utils.calculate_f1 == correct[0] / correct[1]; the gate compares two such counts. */
int sgs_masked_correct(const float* logits, int64_t N, int64_t C, const int64_t* y, const uint8_t* train_mask,
                       int32_t* correct, sgs_stream_t stream);
/* The F1 gate's two counts (learned vs random logits, training_hybrid.py:92-101) in one launch:
 * correct4 = {#correct_a, #train, #correct_b, #train}; correct4 must be ZERO on entry (it is accumulated into). */
int sgs_masked_correct_pair(const float* logits_a, const float* logits_b, int64_t N, int64_t C, const int64_t* y,
                            const uint8_t* train_mask, int32_t* correct4, sgs_stream_t stream);

/* The gate of training_hybrid.py:95-103 in two launches, no zero fill, no atomics: out5[0..3] = (#correct_a, #train, #correct_b, #train)
 * (argmax of each logit matrix against y on the train rows; lowest index on ties), out5[4] = 0.  With dst_host_mapped (pinned,
 * device-mapped host int32[5]) the finishing launch also hands the four counts to the host as sgs_publish_to_host does (payload, then
 * seq_dev[0] -- or 1 -- as the sequence word with release semantics).  ws: sgs_gate_counts_workspace_bytes(N). */
size_t sgs_gate_counts_workspace_bytes(int64_t N);
int sgs_gate_counts(const float* logits_a, const float* logits_b, int64_t N, int64_t C, const int64_t* y, const uint8_t* train_mask,
                    int32_t* out5, const uint64_t* seq_dev, int32_t* dst_host_mapped, void* ws, size_t ws_bytes, sgs_stream_t stream);

/* Closing launch of a replayed step: loss_sum[0] += loss[0] and epoch[0] += 1 (either pair may be NULL). */
int sgs_loss_tick(float* loss_sum, const float* loss, uint64_t* epoch, sgs_stream_t stream);
/* Publish n (<= 63) device words to pinned, device-mapped HOST memory: dst[0..n) = src[0..n), then dst[n] = low 32 bits
 * of *seq_dev (NULL: 1) with release semantics at system scope.  A host thread polling dst[n] for a change reads the
 * payload without a copy-engine round trip or a stream synchronisation (the gate read-back of a replayed step). */
int sgs_publish_to_host(const int32_t* src_dev, int64_t n, const uint64_t* seq_dev, int32_t* dst_host_mapped, sgs_stream_t stream);

/* criterion(out[train_mask], y[train_mask]) for criterion = nn.CrossEntropyLoss() (main.py:125;
 * training_hybrid.py:105,139,145): loss[0] = mean over train rows of (logsumexp - logit[y]).
 * row_lse[N], rowloss[N], n_rows[1] are caller scratch kept for backward:
 * dlogits[i,c] = (softmax - onehot) * grad_loss / #train on train rows, 0 elsewhere. */
int sgs_masked_ce_fwd(const float* logits, int64_t N, int64_t C, const int64_t* y, const uint8_t* train_mask,
                      float* loss, float* row_lse, float* rowloss, int32_t* n_rows, sgs_stream_t stream);
int sgs_masked_ce_bwd(const float* logits, int64_t N, int64_t C, const int64_t* y, const uint8_t* train_mask,
                      const float* row_lse, const int32_t* n_rows, const float* grad_loss, float* dlogits,
                      sgs_stream_t stream);

/* The two edge regularisers on the q sampled edges (weights w, endpoints sampled_edge_index [2,q]):
 *   reg1 = BCE(w[valid], [y_s == y_d]), valid = both endpoints are train nodes; 0 unless the labels sum
 *          to more than 1 (training_hybrid.py:107-129; the reference's torch.isin + .item() sync)
 *   reg2 = mean_j (w_j - cos(logits[s_j], logits[d_j]))^2           (utils.consistency_loss)
 * out[5] = {reg1, reg2, #valid, sum labels, coef1*reg1 + coef2*reg2}.
 * Backward: dw[q] and per-edge gradient rows Gs, Gd [q,C] wrt logits[src], logits[dst]; the caller
 * folds them into dlogits with sgs_endpoint_reduce(Gs, Gd, NULL, +1, +1) over the sampled graph. */
size_t sgs_edge_reg_workspace_bytes(int64_t q);
int sgs_edge_reg_fwd(const float* w, const int64_t* sampled_edge_index, int64_t q, const float* logits, int64_t N,
                     int64_t C, const int64_t* y, const uint8_t* train_mask, float coef1, float coef2, float* out,
                     float* cos_out, void* ws, size_t ws_bytes, sgs_stream_t stream);
int sgs_edge_reg_bwd(const float* w, const int64_t* sampled_edge_index, int64_t q, int64_t q_global, const float* logits,
                     int64_t N, int64_t C, const int64_t* y, const uint8_t* train_mask, const float* out, float coef1,
                     float coef2, const float* grad_loss, float* dw, float* Gs, float* Gd, sgs_stream_t stream);

/* The learned branch's whole loss (training_hybrid.py:105-133: criterion + coef1 reg1 + coef2 reg2) in three launches, for
 * criterion = nn.CrossEntropyLoss():  out[7] = {reg1, reg2, #valid, sum labels, coef1 reg1 + coef2 reg2, cross entropy, loss};
 * row_lse[N], rowloss[N], n_rows[1] as sgs_masked_ce_fwd.  Backward: sgs_edge_reg_bwd (reads out[0..4]) -> sgs_endpoint_reduce ->
 * sgs_masked_ce_bwd_acc, which ADDS the cross entropy's gradient to the dlogits already there. */
int sgs_hybrid_loss_fwd(const float* logits, int64_t N, int64_t C, const int64_t* y, const uint8_t* train_mask, const float* w,
                        const int64_t* sampled_edge_index, int64_t q, float coef1, float coef2, float* out, float* row_lse, float* rowloss,
                        int32_t* n_rows, void* ws, size_t ws_bytes, sgs_stream_t stream);
int sgs_masked_ce_bwd_acc(const float* logits, int64_t N, int64_t C, const int64_t* y, const uint8_t* train_mask, const float* row_lse,
                          const int32_t* n_rows, const float* grad_loss, float* dlogits, sgs_stream_t stream);

/* Edge-sharded losses: raw[4] = {sum bce, sum (w-cos)^2, #valid, sum labels} over THIS rank's sampled edges; the
 * ranks all-reduce raw, form reg1 / reg2 with the global q, and call sgs_edge_reg_bwd with out[2], out[3] = the
 * global #valid / label sum and q_global = the global number of sampled edges (q_global = q when unsharded). */
int sgs_edge_reg_partial(const float* w, const int64_t* sampled_edge_index, int64_t q, const float* logits, int64_t N,
                         int64_t C, const int64_t* y, const uint8_t* train_mask, float* raw, void* ws, size_t ws_bytes,
                         sgs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K8: GAT attention (PyG 2.3.1 GATConv, heads = 1; model.py:189-208 via torch_geometric's GAT):
 *   e_k = leaky_relu(a_src[src_k] + a_dst[dst_k], slope) over each node's in-edges + one self loop
 *   (existing (i,i) edges are ignored, as PyG removes them), soft = softmax per destination
 *   (denominator + 1e-16), alpha = dropout(soft, p) keyed by (seed, site, edge id) / (site+1, node).
 * sgs_gat_alpha_fwd writes soft/alpha in dst-CSR entry order (+ per-node loop values); the aggregation
 * out = sum_k alpha_k x'[src_k] + alpha_loop x'[i] + bias is sgs_spmm_csr(val = alpha_in, diag =
 * alpha_loop); backward: galpha (per edge id) / gloop from sgs_sddmm_csr, then sgs_gat_alpha_bwd gives
 * g_edge[eid] = dL/d(a_src[src]+a_dst[dst]) per edge, g_selfloop[i], and d_a_dst[i]; d_a_src is the
 * per-source sum of g_edge (sgs_spmm_csr over the src-CSR with D = 1) + g_selfloop.
 * sgs_gather_by_eid / sgs_scatter_by_eid re-order per-edge arrays between edge-id and CSR entry order.
 * ---------------------------------------------------------------------------------- */
/* Node-level attention scores of a GATConv layer in one pass over x' [N, D]: a_src[i] = <x'[i, :], att_src>, a_dst likewise (GATConv's
 * (x' * att).sum(-1)); backward: dxl (+)= g_src (x) att_src + g_dst (x) att_dst (accumulate != 0: added to dxl), d att_src = sum_i g_src[i] x'[i, :],
 * d att_dst likewise (fixed summation order).  ws: sgs_gat_scores_bwd_workspace_bytes(N, D). */
int sgs_gat_scores_fwd(const float* xl, int64_t N, int64_t D, const float* att_src, const float* att_dst, float* a_src, float* a_dst,
                       sgs_stream_t stream);
size_t sgs_gat_scores_bwd_workspace_bytes(int64_t N, int64_t D);
int sgs_gat_scores_bwd(const float* xl, int64_t N, int64_t D, const float* att_src, const float* att_dst, const float* g_src, const float* g_dst,
                       int accumulate, float* dxl, float* datt_src, float* datt_dst, void* ws, size_t ws_bytes, sgs_stream_t stream);
int sgs_gat_alpha_fwd(const float* a_src, const float* a_dst, int64_t N, int64_t n_edges, const int32_t* in_ptr,
                      const int32_t* in_src, const int32_t* in_eid, float negative_slope, float p_drop, uint64_t seed,
                      uint32_t site, float* soft_in, float* soft_loop, float* alpha_in, float* alpha_loop,
                      sgs_stream_t stream);
int sgs_gat_alpha_bwd(const float* a_src, const float* a_dst, int64_t N, int64_t n_edges, const int32_t* in_ptr,
                      const int32_t* in_src, const int32_t* in_eid, float negative_slope, float p_drop, uint64_t seed,
                      uint32_t site, const float* soft_in, const float* soft_loop, const float* galpha,
                      const float* gloop, float* g_edge, float* g_selfloop, float* d_a_dst, sgs_stream_t stream);
int sgs_gather_by_eid(const float* by_eid, const int32_t* eid, int64_t n, float* out_order, sgs_stream_t stream);
int sgs_scatter_by_eid(const float* in_order, const int32_t* eid, int64_t n, float* by_eid, sgs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Weight-gradient GEMM of the node-level Linear layers: C[M,N] = A^T B, A [K,M], B [K,N] row-major,
 * K = number of graph nodes (dW = dY^T X for GCNConv.lin, model.py:94-95,151-153).  fp32 MFMA fed from
 * coalesced global reads, split-K with a fixed-order combine (deterministic).  Skinny shapes only (the
 * vendor GEMM serves the rest): meant for M, N <= ~1k.  K >= 8192 (with M % 4 == 0, N % 2 == 0) takes a tall-K kernel:
 * 128 x 64 output tile and a K-slice per wave, one 16-B + one 8-B load per 8 MFMAs (dW1a = dv^T feat, K = q rows).
 * ---------------------------------------------------------------------------------- */
size_t sgs_gemm_tn_workspace_bytes(int64_t K, int64_t M, int64_t N);
int sgs_gemm_tn(const float* A, const float* B, int64_t K, int64_t M, int64_t N, float* C, void* ws, size_t ws_bytes,
                sgs_stream_t stream);
/* Same product with the column sums of A as a by-product (colsum_A [M]; d b1 = colsum(dv) of the scorer's backward rides
 * on d W1a = dv^T feat): only for the shapes the tall-K kernel serves with a split -- ask sgs_gemm_tn_can_colsum first. */
void sgs_gemm_tn_set_tall_variant(int variant);   /* tall-K shapes (K >= 8192): 1 / -1 = bf16x6 kernel (default; fp32-faithful, see sgs_edge_score_set_variant), 0 = fp32-MFMA kernel */
int sgs_gemm_tn_can_colsum(int64_t K, int64_t M, int64_t N);
int sgs_gemm_tn_colsum(const float* A, const float* B, int64_t K, int64_t M, int64_t N, float* C, float* colsum_A, void* ws,
                       size_t ws_bytes, sgs_stream_t stream);
/* The same product written with row stride ldc >= N, i.e. into a column block of a wider matrix: the two halves of d fc1.weight
 * [H, 2H] (model.py:29-31: W1 [x*y | x-y]) are d W1a = dv^T feat and d W1b = dU^T codes, each an [H, H] block with ldc = 2H.
 * colsum_A may be NULL (otherwise as sgs_gemm_tn_colsum).  Workspace: sgs_gemm_tn_workspace_bytes(K, M, N). */
int sgs_gemm_tn_ld(const float* A, const float* B, int64_t K, int64_t M, int64_t N, float* C, int64_t ldc, float* colsum_A, void* ws,
                   size_t ws_bytes, sgs_stream_t stream);

/* ----------------------------------------------------------------------------------
 * Effective-resistance edge prior (datasets.py:159-173 add_ER; estimator: EffectiveResistanceWeights.ipynb cell 11 er_edge).
 * weight[e] = max(0, sum_{i < walk_lengths} (X_is/deg s - X_it/deg t - Y_is/deg s + Y_it/deg t) / walks) with X / Y the
 * numbers of `walks` uniform random walks of length i from s / t that end at s or t (reference: walk_lengths = 4, walks = 100).
 * out_ptr / out_dst: CSR of the symmetric, coalesced edge list (sgs_graph_build); counter-based randomness keyed on
 * (seed, edge, walk, step).  The host applies softmax(weight * E^-1/2) as add_ER does.
 * ---------------------------------------------------------------------------------- */
int sgs_er_weight(const int64_t* edge_index, int64_t E, int64_t N, const int32_t* out_ptr, const int32_t* out_dst, int walk_lengths,
                  int walks, uint64_t seed, float* weight, sgs_stream_t stream);

/* ----------------------------------------------------------------------------------
 * Optimiser (training_hybrid.py:22-27, 135-141: two torch.optim.Adam steps per batch).
 * One launch updates up to sgs_adam_max_tensors() tensors with torch.optim.Adam's rule (coupled weight decay, no amsgrad):
 *   desc_host [n_tensors][7] int64 in HOST memory, read during the call only (the descriptors travel by value in the
 *             kernel arguments): {param, grad, exp_avg, exp_avg_sq, numel, step, gate}; the pointers are DEVICE addresses;
 *             `step` is that tensor's float counter of completed steps (one per parameter, as torch), read and then
 *             incremented by the kernel, so the call is capturable into a HIP graph; `gate` is 0 or the address of a
 *             device float: while it reads 0 the tensor (and its counter) is left untouched (data-parallel training: the
 *             scorer's tensors are skipped on steps where no rank took the learned branch, decided without a host round trip)
 *   ticket    device uint32, zero on first use (the kernel leaves it zero); one per concurrently running call
 * ---------------------------------------------------------------------------------- */
int sgs_adam_max_tensors(void);
int sgs_adam_step(const int64_t* desc_host, int64_t n_tensors, float lr, float beta1, float beta2, float eps, float weight_decay,
                  int maximize, uint32_t* ticket, sgs_stream_t stream);
/* Several optimisers' steps in ONE launch, plus the step's closing bookkeeping (round 3).  training_hybrid.py:136-137 steps
 * optimizer_edge_prob and then optimizer_gnn; the two overlap on edge_prob_mlp.gcn* (main.py:100-109, 122), so those tensors are updated
 * twice with the same gradient.  A descriptor names a tensor once and carries one or two optimiser states, applied in order:
 *   desc_host  [n_tensors][11] int64: {param, grad, numel, exp_avg, exp_avg_sq, step, gate, exp_avg2, exp_avg_sq2, step2, gate2}
 *              (exp_avg2 == 0: one state; gates as in sgs_adam_step)
 *   hyper_host [n_tensors][12] float: {lr, beta1, beta2, eps, weight_decay, maximize} of the first state, then of the second
 * loss_sum / loss / epoch (each pair optional): the last workgroup also does what sgs_loss_tick does.  n_tensors <= sgs_adam_multi_max_tensors(). */
int sgs_adam_multi_max_tensors(void);
int sgs_adam_step_multi(const int64_t* desc_host, const float* hyper_host, int64_t n_tensors, uint32_t* ticket, float* loss_sum, const float* loss,
                        uint64_t* epoch, sgs_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SGS_HIP_H_ */
