"""Edge-partitioned execution of one graph over R ranks (BASELINE.json config 5).

Node data (x, y, masks, parameters) are replicated; the EDGE list is split into R contiguous shards
whose boundaries are multiples of the sampler chunk (2048).  Per rank and step:

  * scoring is local: every rank scores only its own edges (no communication);
  * draws are one exact GLOBAL exponential-race top-q (`dist_sample_topq`): per-chunk partial sums are
    all-gathered and reduced in the single-GPU order (bit-identical normaliser), the three 2048-bin
    digit histograms are all-reduced, ties at the threshold go to the lowest global edge ids, and the
    compaction is local.  Noise is keyed by the global edge id, so the selected set does not depend on
    the number of ranks;
  * GCN layers aggregate each rank's edges into a partial [N, D] sum and ALL-REDUCE it (the "RCCL
    all-reduce of node embeddings over xGMI" of the north star); degrees are all-reduced once per graph;
    bias / ReLU / dropout run after the all-reduce on every rank (replicated).

Scope this round: the forward / inference path (`sharded_evaluate_forward`, i.e. evaluate.py's learned
mode on a graph too large for one GPU).  The training backward needs the matching gradient all-reduces
(Megatron-style f/g operators at the shard boundaries) and is not built yet.
Collectives go through torch.distributed: backend "nccl" is RCCL on ROCm; tests use gloo.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import _lib, ops
from .model import SITE_ENC, SITE_GNN, SITE_SCORE


def _world():
    return (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)


def shard_bounds(E_total: int, world: int, chunk: int = 2048):
    """Contiguous edge ranges, boundaries at multiples of `chunk` (the sampler's reduction granule)."""
    nchunks = (E_total + chunk - 1) // chunk
    per = (nchunks + world - 1) // world
    b = [min(r * per * chunk, E_total) for r in range(world + 1)]
    b[-1] = E_total
    return b


class EdgeShard:
    """Replicated node data + this rank's slice of the edge list."""

    def __init__(self, batch, rank: int, world: int):
        E = batch.edge_index.shape[1]
        b = shard_bounds(E, world, int(_lib.lib().sgs_sampler_chunk()))
        lo, hi = b[rank], b[rank + 1]
        self.rank, self.world, self.E_total, self.edge_offset, self.bounds = rank, world, E, lo, b
        self.x, self.y = batch.x, batch.y
        self.train_mask, self.val_mask, self.test_mask = batch.train_mask, getattr(batch, "val_mask", None), getattr(batch, "test_mask", None)
        self.edge_index = batch.edge_index[:, lo:hi].contiguous()
        self.prob = batch.prob[lo:hi].contiguous() if getattr(batch, "prob", None) is not None else None
        self.N = batch.x.shape[0]


def _all_gather_concat(t: torch.Tensor, sizes):
    """All-gather 1-D tensors of per-rank lengths `sizes` and concatenate them in rank order."""
    rank, world = _world()
    if world == 1:
        return t
    m = max(max(sizes), 1)
    buf = torch.zeros(m, dtype=t.dtype, device=t.device)
    buf[:t.numel()] = t
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    return torch.cat([o[:n] for o, n in zip(out, sizes)])


def dist_sample_topq(mode: int, p_local, prior_local, c: float, q: int, edge_index_local, edge_offset: int, bounds,
                     noise_local=None, seed: int = 0, stream_id: int = 0, want_keys: bool = False) -> ops.SampleResult:
    """One exact global top-q draw over edge-sharded keys.  Returns this rank's part: mask over the local
    edges, GLOBAL ids of the selected local edges (ascending), their columns and probabilities, and
    stats = {Z, max, threshold key, #ties taken} (identical on every rank)."""
    L = _lib.lib()
    rank, world = _world()
    dev = p_local.device
    ops._need_gpu(p_local, prior_local, edge_index_local, noise_local)
    chunk = int(L.sgs_sampler_chunk())
    E = p_local.numel()
    E_total = bounds[-1]
    if q > E_total:
        raise RuntimeError(f"cannot sample q={q} > E={E_total} edges without replacement")
    nblk = [(bounds[r + 1] - bounds[r] + chunk - 1) // chunk for r in range(world)]
    st = ops._stream()
    p_local = p_local.detach().contiguous()
    ws = torch.empty(int(L.sgs_sampler_shard_workspace_bytes(E)), dtype=torch.uint8, device=dev)     # private: lives across phases
    scal = torch.zeros(4, dtype=torch.float32, device=dev)
    part = torch.zeros(max(nblk[rank], 1), dtype=torch.float32, device=dev)

    def reduce_stage(stage):
        _lib.check(L.sgs_sampler_shard_partials(stage, ops._ptr(p_local), E, ops._ptr(scal), ops._ptr(part), st), "shard_partials")
        allp = _all_gather_concat(part[:nblk[rank]], nblk).contiguous()
        _lib.check(L.sgs_sampler_shard_finalize(stage, ops._ptr(allp), allp.numel(), ops._ptr(scal), st), "shard_finalize")

    if mode == ops.SAMPLE_LEARNED:
        reduce_stage(0)
    else:
        reduce_stage(1)
        reduce_stage(2)

    r = ops.SampleResult()
    r.E, r.q = E, q
    r.keys = torch.empty(E, dtype=torch.float32, device=dev) if want_keys else None
    hist = torch.zeros(2048, dtype=torch.int32, device=dev)
    state = torch.zeros(4, dtype=torch.int32, device=dev)
    _lib.check(L.sgs_sampler_shard_keys(mode, ops._ptr(p_local), ops._ptr(prior_local), float(c), ops._ptr(noise_local), seed,
                                        stream_id, edge_offset, E, ops._ptr(scal), ws.data_ptr(), ops._ptr(r.keys), ops._ptr(hist), st),
               "shard_keys")
    for ps in range(3):
        if ps > 0:
            _lib.check(L.sgs_sampler_shard_hist(ws.data_ptr(), E, ps, ops._ptr(state), ops._ptr(hist), st), "shard_hist")
        if world > 1:
            dist.all_reduce(hist)                                      # integer counts: exact
        _lib.check(L.sgs_sampler_select(ops._ptr(hist), ps, q, ops._ptr(state), st), "sampler_select")   # zeroes hist
    counts = torch.zeros(2, dtype=torch.int32, device=dev)
    _lib.check(L.sgs_sampler_shard_count(ws.data_ptr(), E, ops._ptr(state), ops._ptr(counts), ws.data_ptr(), ws.numel(), st), "shard_count")
    if world > 1:
        allc = [torch.empty_like(counts) for _ in range(world)]
        dist.all_gather(allc, counts)
        allc = torch.stack(allc).tolist()
    else:
        allc = [counts.tolist()]
    k_rem = int(state[1].item())                                       # ties to take globally, lowest ids first
    ties = []
    left = k_rem
    for g_, e_ in allc:
        t_ = min(left, e_)
        ties.append(t_)
        left -= t_
    q_local = allc[rank][0] + ties[rank]
    r.mask = torch.empty(E, dtype=torch.bool, device=dev)
    r.eid = torch.empty(q_local, dtype=torch.int64, device=dev)
    r.edge_index = torch.empty(2, q_local, dtype=torch.int64, device=dev)
    r.p = torch.empty(q_local, dtype=torch.float32, device=dev)
    _lib.check(L.sgs_sampler_shard_compact(ws.data_ptr(), E, ops._ptr(state), ties[rank], q_local, edge_offset, ops._ptr(p_local),
                                           ops._ptr(edge_index_local), ops._ptr(r.mask), ops._ptr(r.eid), ops._ptr(r.edge_index),
                                           ops._ptr(r.p), ws.data_ptr(), ws.numel(), st), "shard_compact")
    r.stats = torch.stack([scal[0], scal[1], state[0:1].view(torch.float32)[0], torch.tensor(float(k_rem), device=dev)])
    return r


# ------------------------------------------------------------------ edge-sharded GCN forward
def sharded_norm(graph: ops.Graph, w):
    """gcn_norm over the union of all ranks' edges: all-reduce of the per-rank in-degree sums."""
    L = _lib.lib()
    _, world = _world()
    dev = graph.edge_index.device
    f32 = dict(dtype=torch.float32, device=dev)
    deg = torch.empty(graph.N, **f32)
    _lib.check(L.sgs_gcn_degree_partial(ops._ptr(w), graph.n_edges, graph.N, ops._ptr(graph.in_ptr), ops._ptr(graph.in_src),
                                        ops._ptr(graph.in_eid), ops._ptr(deg), ops._stream()), "sgs_gcn_degree_partial")
    if world > 1:
        dist.all_reduce(deg)
    nm = ops.Norm()
    nm.graph, nm.w, nm.handle = graph, w, None
    nm.dis, nm.loopw, nm.what_loop = torch.empty(graph.N, **f32), torch.empty(graph.N, **f32), torch.empty(graph.N, **f32)
    nm.what_in, nm.what_out = torch.empty(max(graph.n_edges, 1), **f32), torch.empty(max(graph.n_edges, 1), **f32)
    _lib.check(L.sgs_gcn_norm_from_degree(ops._ptr(w), ops._ptr(deg), graph.n_edges, graph.N, ops._ptr(graph.in_ptr),
                                          ops._ptr(graph.in_src), ops._ptr(graph.in_eid), ops._ptr(graph.out_ptr),
                                          ops._ptr(graph.out_dst), ops._ptr(graph.out_eid), ops._ptr(nm.dis), ops._ptr(nm.loopw),
                                          ops._ptr(nm.what_in), ops._ptr(nm.what_out), ops._ptr(nm.what_loop), ops._stream()),
               "sgs_gcn_norm_from_degree")
    return nm


def sharded_propagate(xl, nm, bias, act=ops.ACT_NONE, p=0.0, seed=0, site=0):
    """act( all_reduce_r( A_hat_r xl ) + bias ): partial aggregate over this rank's edges, RCCL all-reduce of the
    [N, D] node embeddings, replicated epilogue.  The self-loop term is added by rank 0 only."""
    L = _lib.lib()
    rank, world = _world()
    gr = nm.graph
    N, D = xl.shape
    part = ops._spmm(xl.contiguous(), gr.in_ptr, gr.in_src, nm.what_in, nm.what_loop if rank == 0 else None, None, ops.ACT_NONE, 0.0,
                     0, 0, N, D, gr.n_edges)
    if world > 1:
        dist.all_reduce(part)
    Y = torch.empty_like(part)
    _lib.check(L.sgs_bias_act(ops._ptr(part), ops._ptr(bias), N, D, act, float(p), seed, site, ops._ptr(Y), ops._stream()), "sgs_bias_act")
    return Y


@torch.no_grad()
def sharded_evaluate_forward(args, model, shard: EdgeShard, q: int, noise_local=None, seed: int = 0, stream_id: int = 1):
    """evaluate.py:14-20 (mode 'learned', model.eval()) on an edge-sharded graph: EdgeProbGCN encoder over
    all E edges, scores for the local edges, one global istest draw, weighted 2-layer GCN -> logits
    (replicated on every rank).  Returns (logits, local SampleResult)."""
    sc = model.edge_prob_mlp
    x, N = shard.x, shard.N
    H = sc.fc1.weight.shape[0]
    g_full = ops.get_graph(shard.edge_index, N)
    nm = sharded_norm(g_full, None)
    h = sharded_propagate(x @ sc.gcn1.lin.weight.t(), nm, sc.gcn1.bias, act=ops.ACT_RELU)
    codes = sharded_propagate(h @ sc.gcn2.lin.weight.t(), nm, sc.gcn2.bias, act=ops.ACT_RELU)
    p_local = ops.edge_score(codes, sc.fc1.weight, sc.fc1.bias, sc.fc2.weight, sc.fc2.bias, shard.edge_index)
    smp = dist_sample_topq(ops.SAMPLE_LEARNED, p_local, None, args.degree_bias_coef, q, shard.edge_index, shard.edge_offset,
                           shard.bounds, noise_local=noise_local, seed=seed, stream_id=stream_id)
    local_ids = smp.eid - shard.edge_offset
    w = ops.st_weights(p_local, None, args.degree_bias_coef, smp.stats, local_ids)        # sampling.py:137-155 (Z is global)
    g_s = ops.get_graph(smp.edge_index, N)
    nms = sharded_norm(g_s, w)
    h1 = sharded_propagate(x @ model.gcn1.lin.weight.t(), nms, model.gcn1.bias, act=ops.ACT_RELU)
    out = sharded_propagate(h1 @ model.gcn2.lin.weight.t(), nms, model.gcn2.bias)
    return out, smp
