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

Built: the inference path (`sharded_evaluate_forward` = evaluate.py's learned mode) and the hybrid TRAINING
step (`train_step_sharded`): replicated tensors enter sharded compute through `_F` (identity forward,
all-reduce backward) and partial results leave through `_G` (all-reduce forward, identity backward), the
gcn_norm backward's per-node term and the regulariser sums are all-reduced, so every rank finishes backward
with complete, identical gradients and no separate gradient synchronisation is needed.
Collectives go through torch.distributed: backend "nccl" is RCCL on ROCm; tests use gloo (R processes on one GPU).
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import _lib, ops
from .model import SITE_ENC, SITE_GNN, SITE_SCORE


def _world():
    return (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)


def _comm() -> bool:
    """Are the collectives issued?  More than one rank -- or SGS_DP_FORCE=1 with an initialised group, which lets a one-GPU box run
    every collective of the sharded step through RCCL (world size 1)."""
    from .dist import is_parallel
    return is_parallel()


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
        if self.edge_index.is_cuda and self.edge_index.shape[1] >= 65536:
            # mates INSIDE the shard (a contiguous range of source nodes holds both directions of the edges between its own nodes):
            # the paired scorer forward halves the contraction work for those, every other edge runs alone -- same p either way
            ops.get_pairs(self.edge_index, self.N, build=True)


def _all_gather_concat(t: torch.Tensor, sizes):
    """All-gather 1-D tensors of per-rank lengths `sizes` and concatenate them in rank order."""
    rank, world = _world()
    if not _comm():
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
        if _comm():
            dist.all_reduce(hist)                                      # integer counts: exact
        _lib.check(L.sgs_sampler_select(ops._ptr(hist), ps, q, ops._ptr(state), st), "sampler_select")   # zeroes hist
    counts = torch.zeros(2, dtype=torch.int32, device=dev)
    _lib.check(L.sgs_sampler_shard_count(ws.data_ptr(), E, ops._ptr(state), ops._ptr(counts), ws.data_ptr(), ws.numel(), st), "shard_count")
    if _comm():
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
    if _comm():
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
    if _comm():
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
    g_s = ops.get_subgraph(shard.edge_index, N, smp, eid=local_ids)          # squeezed out of the shard's cached CSR
    nms = sharded_norm(g_s, w)
    h1 = sharded_propagate(x @ model.gcn1.lin.weight.t(), nms, model.gcn1.bias, act=ops.ACT_RELU)
    out = sharded_propagate(h1 @ model.gcn2.lin.weight.t(), nms, model.gcn2.bias)
    return out, smp


# ======================================================================================================
# Edge-sharded TRAINING step (hybrid pipeline).  Replicated tensors enter sharded compute through `_F`
# (identity forward, all-reduce backward) and partial results leave it through `_G` (all-reduce forward,
# identity backward), so every rank ends the backward pass with complete, identical parameter gradients
# and the replicas stay in lock-step without a separate gradient synchronisation.
# ======================================================================================================
class _F(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        if _comm():
            g = g.clone()
            dist.all_reduce(g)
        return g


class _G(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = x.clone()
        if _comm():
            dist.all_reduce(y)
        return y

    @staticmethod
    def backward(ctx, g):
        return g


class _BiasAct(torch.autograd.Function):
    """Replicated layer epilogue after the embedding all-reduce: Y = act(X + bias)."""

    @staticmethod
    def forward(ctx, X, bias, act, p, seed, site):
        L = _lib.lib()
        N, D = X.shape
        Y = torch.empty_like(X)
        _lib.check(L.sgs_bias_act(ops._ptr(X.contiguous()), ops._ptr(bias), N, D, act, float(p), seed, site, ops._ptr(Y), ops._stream()),
                   "sgs_bias_act")
        ctx.act, ctx.p, ctx.has_bias = act, float(p), bias is not None
        ctx.save_for_backward(Y if act != ops.ACT_NONE else None)
        return Y

    @staticmethod
    def backward(ctx, dY):
        L = _lib.lib()
        (Y,) = ctx.saved_tensors
        dY = dY.contiguous()
        if ctx.act != ops.ACT_NONE:
            dZ = torch.empty_like(dY)
            _lib.check(L.sgs_act_bwd(ops._ptr(dY), ops._ptr(Y), dY.numel(), ctx.act, ctx.p, ops._ptr(dZ), ops._stream()), "sgs_act_bwd")
        else:
            dZ = dY
        return dZ, (ops._colsum(dZ) if ctx.has_bias else None), None, None, None, None


class _ShardedNorm(torch.autograd.Function):
    """gcn_norm over the union of all ranks' edges, differentiable wrt this rank's edge weights."""

    @staticmethod
    def forward(ctx, w, graph, box):
        nm = sharded_norm(graph, w)
        if _world()[0] != 0:
            nm.what_loop = None                      # the self-loop term is added (and differentiated) on rank 0 only
        box.append(nm)
        ctx.nm = nm
        return torch.empty(graph.n_edges + graph.N, dtype=torch.float32, device=w.device)

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        nm, gr = ctx.nm, ctx.nm.graph
        rank, world = _world()
        g = g.contiguous()
        n, N = gr.n_edges, gr.N
        gw = g[:n].contiguous() if n > 0 else torch.zeros(1, dtype=torch.float32, device=g.device)
        gl = g[n:].contiguous() if rank == 0 else torch.zeros(N, dtype=torch.float32, device=g.device)
        Hn = torch.empty(N, dtype=torch.float32, device=g.device)
        wv = nm.w if n > 0 else torch.zeros(1, dtype=torch.float32, device=g.device)
        _lib.check(L.sgs_gcn_norm_bwd_node(ops._ptr(wv), ops._ptr(gw), ops._ptr(gl), n, N, ops._ptr(nm.dis), ops._ptr(nm.loopw),
                                           ops._ptr(gr.in_ptr), ops._ptr(gr.in_src), ops._ptr(gr.in_eid), ops._ptr(gr.out_ptr),
                                           ops._ptr(gr.out_dst), ops._ptr(gr.out_eid), ops._ptr(Hn), ops._stream()), "sgs_gcn_norm_bwd_node")
        if _comm():
            dist.all_reduce(Hn)
        dw = torch.empty(n, dtype=torch.float32, device=g.device)
        if n > 0:
            _lib.check(L.sgs_gcn_norm_bwd_edge(ops._ptr(gw), ops._ptr(gl), n, N, ops._ptr(nm.dis), ops._ptr(gr.loop_eid),
                                               ops._ptr(gr.edge_index), ops._ptr(Hn), ops._ptr(dw), ops._stream()), "sgs_gcn_norm_bwd_edge")
        return dw, None, None


def sharded_norm_autograd(graph, w):
    rank, _ = _world()
    if w is None or not (w.requires_grad and torch.is_grad_enabled()):
        nm = sharded_norm(graph, None if w is None else w.detach().contiguous())
        if rank != 0:
            nm.what_loop = None
        return nm
    box = []
    handle = _ShardedNorm.apply(w.contiguous(), graph, box)
    nm = box[0]
    nm.handle = handle
    return nm


def sharded_gcn_layer(x, W, bias, nm, act=ops.ACT_NONE, p=0.0, seed=0, site=0):
    """One GCN layer on an edge-sharded graph with autograd: replicated X W^T -> f -> local aggregate -> g -> epilogue."""
    xl = _F.apply(ops.linear_nobias(x, W))
    part = ops._Propagate.apply(xl.contiguous(), nm.handle, None, nm, ops.ACT_NONE, 0.0, 0, 0)
    return _BiasAct.apply(_G.apply(part), bias, act, float(p), int(seed), int(site))


class _ShardedEdgeReg(torch.autograd.Function):
    """coef1 * reg1 + coef2 * reg2 over the union of all ranks' sampled edges."""

    @staticmethod
    def forward(ctx, w, logits, sei, y, mask_u8, graph, coef1, coef2, q_global):
        L = _lib.lib()
        q = w.numel()
        N, C = logits.shape
        dev = w.device
        raw = torch.empty(4, dtype=torch.float32, device=dev)
        ws = ops.workspace(L.sgs_edge_reg_workspace_bytes(q), dev)
        _lib.check(L.sgs_edge_reg_partial(ops._ptr(w), ops._ptr(sei), q, ops._ptr(logits), N, C, ops._ptr(y), ops._ptr(mask_u8),
                                          ops._ptr(raw), ws.data_ptr(), ws.numel(), ops._stream()), "sgs_edge_reg_partial")
        if _comm():
            dist.all_reduce(raw)
        reg1 = torch.where(raw[3] > 1.0, raw[0] / raw[2], torch.zeros((), device=dev))          # training_hybrid.py:125-128
        reg2 = raw[1] / float(q_global)
        out = torch.stack([reg1, reg2, raw[2], raw[3], coef1 * reg1 + coef2 * reg2]).contiguous()
        ctx.save_for_backward(w, logits, sei, y, mask_u8, out)
        ctx.graph, ctx.coef1, ctx.coef2, ctx.q_global = graph, float(coef1), float(coef2), int(q_global)
        return out[4].clone()

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        w, logits, sei, y, mask_u8, out = ctx.saved_tensors
        q = w.numel()
        N, C = logits.shape
        dev = w.device
        g = g.reshape(1).contiguous().float()
        dw = torch.empty(q, dtype=torch.float32, device=dev)
        Gs = torch.empty(max(q, 1), C, dtype=torch.float32, device=dev)
        Gd = torch.empty(max(q, 1), C, dtype=torch.float32, device=dev)
        _lib.check(L.sgs_edge_reg_bwd(ops._ptr(w), ops._ptr(sei), q, ctx.q_global, ops._ptr(logits), N, C, ops._ptr(y), ops._ptr(mask_u8),
                                      ops._ptr(out), ctx.coef1, ctx.coef2, ops._ptr(g), ops._ptr(dw), ops._ptr(Gs), ops._ptr(Gd),
                                      ops._stream()), "sgs_edge_reg_bwd")
        dlogits = ops._endpoint_reduce(Gs, Gd, None, ctx.graph, 1.0, 1.0, C)       # partial: the caller's _F all-reduces it
        return dw, dlogits, None, None, None, None, None, None, None


def train_step_sharded(args, model, shard: EdgeShard, optimizer_gnn, optimizer_edge_prob, criterion, q: int, noise=None):
    """One hybrid step (training_hybrid.py:35-141, mode 'learned', E > q, EdgeProbGCN scorer) on an edge-sharded
    graph.  Dropout seeds / noise ticks are consumed in the same order as the single-GPU `train`, rows are
    global ids, so the result matches the unsharded step up to fp32 summation order.  Returns a trace dict."""
    from .model import _DropoutClock
    from .sampling import _NoiseClock
    noise = noise or {}
    model.train()
    optimizer_edge_prob.zero_grad()
    optimizer_gnn.zero_grad()
    sc = model.edge_prob_mlp
    x, N, ei = shard.x, shard.N, shard.edge_index
    p = sc.dropout.p
    act = ops.ACT_RELU_DROPOUT if p > 0 else ops.ACT_RELU
    off, bounds = shard.edge_offset, shard.bounds

    # K0: prior-only draw (global), this rank's random edges
    seed_n, tick = (0, 0) if noise.get("prior") is not None else _NoiseClock.next()
    rs = dist_sample_topq(ops.SAMPLE_PRIOR, shard.prob, None, 0.0, q, ei, off, bounds, noise_local=noise.get("prior"), seed=seed_n,
                          stream_id=tick)
    g_r = ops.get_subgraph(ei, N, rs, eid=rs.eid - off)
    nm_r = sharded_norm_autograd(g_r, None)
    # scorer encoder over the random graph (model.py:106-108)
    h = sharded_gcn_layer(x, sc.gcn1.lin.weight, sc.gcn1.bias, nm_r, act=act, p=p, seed=_DropoutClock.next_seed(), site=SITE_ENC)
    codes = sharded_gcn_layer(h, sc.gcn2.lin.weight, sc.gcn2.bias, nm_r, act=ops.ACT_RELU)
    # scores for this rank's edges; replicated inputs pass through f so their partial gradients are summed
    active = ops.ActiveSet()
    p_local = ops.edge_score(_F.apply(codes), _F.apply(sc.fc1.weight), _F.apply(sc.fc1.bias), _F.apply(sc.fc2.weight),
                             _F.apply(sc.fc2.bias), ei, active=active, p=p, seed=_DropoutClock.next_seed(), site=SITE_SCORE,
                             edge_id_offset=off)
    # K2+K3: learned draw (global)
    seed_n, tick = (0, 0) if noise.get("sample") is not None else _NoiseClock.next()
    smp = dist_sample_topq(ops.SAMPLE_LEARNED, p_local, shard.prob, args.degree_bias_coef, q, ei, off, bounds,
                           noise_local=noise.get("sample"), seed=seed_n, stream_id=tick)
    local_ids = smp.eid - off
    g_s = ops.get_subgraph(ei, N, smp, eid=local_ids)
    active.set(local_ids, g_s)
    w_local = p_local.index_select(0, local_ids)
    nm_s = sharded_norm_autograd(g_s, w_local)
    pg = model.dropout.p
    actg = ops.ACT_RELU_DROPOUT if pg > 0 else ops.ACT_RELU
    h1 = sharded_gcn_layer(x, model.gcn1.lin.weight, model.gcn1.bias, nm_s, act=actg, p=pg, seed=_DropoutClock.next_seed(), site=SITE_GNN)
    learned_out = sharded_gcn_layer(h1, model.gcn2.lin.weight, model.gcn2.bias, nm_s)
    update_edge_mlp, random_out, counts = True, None, None
    if args.conditional:
        h1r = sharded_gcn_layer(x, model.gcn1.lin.weight, model.gcn1.bias, nm_r, act=actg, p=pg, seed=_DropoutClock.next_seed(),
                                site=SITE_GNN)
        random_out = sharded_gcn_layer(h1r, model.gcn2.lin.weight, model.gcn2.bias, nm_r)
        cbuf = torch.empty(4, dtype=torch.int32, device=x.device)
        ops.masked_correct(learned_out, shard.y, shard.train_mask, out=cbuf[0:2])
        ops.masked_correct(random_out, shard.y, shard.train_mask, out=cbuf[2:4])
        counts = cbuf.tolist()
        update_edge_mlp = counts[0] > counts[2]                  # logits are replicated: every rank takes the same branch
    if update_edge_mlp:
        loss = ops.masked_cross_entropy(learned_out, shard.y, shard.train_mask)
        c1 = args.regularizer1_coef if args.reg1 else 0.0
        c2 = args.consist_reg_coef if args.reg2 else 0.0
        if c1 != 0.0 or c2 != 0.0:
            loss = loss + _ShardedEdgeReg.apply(w_local.contiguous(), _F.apply(learned_out).contiguous(), smp.edge_index, shard.y,
                                                ops._u8(shard.train_mask), g_s, float(c1), float(c2), q)
        loss.backward()
        optimizer_edge_prob.step()
        optimizer_gnn.step()
    else:
        loss = ops.masked_cross_entropy(random_out, shard.y, shard.train_mask)
        loss.backward()
        optimizer_gnn.step()
    return dict(loss=loss.detach(), sample=smp, random=rs, update_edge_mlp=update_edge_mlp, learned_out=learned_out.detach(),
                counts=counts)


# ======================================================================================================
# NODE-BLOCK variant of the edge-sharded step: half the embedding bytes of the all-reduce form.
#
# The edge list is row-sorted, so a contiguous edge shard is the OUT-edges of a contiguous range of source nodes: rank r owns the
# node block [nb[r], nb[r+1]) (nb[r] = first source of its shard).  A GCN layer then needs X' only for the rank's own source rows,
# aggregates its edges into a partial [N, D] sum, and the partial sums are REDUCE-SCATTERED into the blocks (forward) -- the bias /
# activation / dropout epilogue and the next layer's X W^T run on 1 / R of the rows -- while backward ALL-GATHERS the block
# gradients.  (This is the mirror image of 8e's "dst-range: all-gather forward, reduce-scatter backward"; with a row-sorted list the
# source side is the contiguous one.  Same volume: (R-1)/R x [N, D] per layer and direction instead of 2 (R-1)/R.)  Full tables are
# rebuilt only where edges gather by arbitrary endpoint: the node codes for the scorer and the logits for the consistency
# regulariser (all-gather forward, reduce-scatter backward).  The one source row a shard boundary may split is exchanged as a
# one-row halo.  Parameter gradients are partial sums on every rank: ONE flat all-reduce at the end of backward.
# Sampling, scoring, gcn_norm and the losses are the same kernels and collectives as in train_step_sharded.
# ======================================================================================================
class NodeBlocks:
    """nb [R+1]: rank r owns nodes nb[r] .. nb[r+1]-1 and needs the source rows nb[r] .. nb[r+1] (the last one is the halo: the
    first node of the next non-empty block, whose edges may start in this shard)."""

    def __init__(self, shard: EdgeShard):
        rank, world = _world()
        self.rank, self.world, self.N = rank, world, shard.N
        ei, dev = shard.edge_index, shard.edge_index.device
        n = ei.shape[1]
        if n > 1 and not bool((ei[0, 1:] >= ei[0, :-1]).all()):
            raise RuntimeError("NodeBlocks: the shard's edge list is not sorted by source (ClusterData / ResidentPartitions emit it sorted)")
        first = ei[0, :1].clone() if n > 0 else torch.full((1,), shard.N, dtype=torch.int64, device=dev)
        if _comm():
            firsts = [torch.empty_like(first) for _ in range(world)]
            dist.all_gather(firsts, first)
            nb = [int(f) for f in firsts]
        else:
            nb = [int(first)]
        nb[0] = 0
        for r in range(len(nb) - 2, -1, -1):              # an empty shard owns nothing: its block collapses onto the next one's start
            nb[r] = min(nb[r], nb[r + 1])
        nb[0] = 0
        self.nb = nb + [shard.N]
        self.lo, self.hi = self.nb[rank], self.nb[rank + 1]
        size = [self.nb[r + 1] - self.nb[r] for r in range(world)]
        # owner of node `hi` = the next rank with a non-empty block (a hub whose edges fill whole shards leaves empty blocks between)
        self.halo_owner = next((r for r in range(rank + 1, world) if size[r] > 0), None) if self.hi < shard.N else None
        self.halo = 0 if self.halo_owner is None else 1
        self.halo_users = [r for r in range(rank) if size[rank] > 0 and self.nb[r + 1] == self.lo
                           and next((k for k in range(r + 1, world) if size[k] > 0), None) == rank]
        self.maxb = max(size)
        n_tot = shard.train_mask.sum().to(torch.int32).reshape(1)          # the mask is replicated: every rank counts the same rows
        self.n_train = n_tot.contiguous()

    @property
    def rows(self):
        return self.hi - self.lo


def node_blocks(shard: EdgeShard) -> NodeBlocks:
    """Cached on the shard (one source-sortedness check and one all-gather of the first sources per shard, not per step)."""
    B = getattr(shard, "_node_blocks", None)
    if B is None:
        B = shard._node_blocks = NodeBlocks(shard)
    return B


def _gather_blocks(block: torch.Tensor, B: NodeBlocks) -> torch.Tensor:
    """[b_r, D] on every rank -> [N, D] (all-gather of padded blocks)."""
    if not _comm():
        return block
    D = block.shape[1]
    buf = torch.zeros(B.maxb, D, dtype=block.dtype, device=block.device)
    buf[:block.shape[0]] = block
    out = [torch.empty_like(buf) for _ in range(B.world)]
    dist.all_gather(out, buf)
    return torch.cat([out[r][:B.nb[r + 1] - B.nb[r]] for r in range(B.world)], dim=0)


def _reduce_scatter_blocks(full: torch.Tensor, B: NodeBlocks) -> torch.Tensor:
    """[N, D] partial sums on every rank -> this rank's block of their sum.  RCCL: reduce_scatter of padded blocks; gloo has no
    reduce_scatter, so the CPU-side tests run all-reduce + slice (same result)."""
    if not _comm():
        return full
    if dist.get_backend() == "nccl":
        D = full.shape[1]
        parts = []
        for r in range(B.world):
            buf = torch.zeros(B.maxb, D, dtype=full.dtype, device=full.device)
            buf[:B.nb[r + 1] - B.nb[r]] = full[B.nb[r]:B.nb[r + 1]]
            parts.append(buf)
        out = torch.empty_like(parts[0])
        dist.reduce_scatter(out, parts)
        return out[:B.rows].contiguous()
    t = full.clone()
    dist.all_reduce(t)
    return t[B.lo:B.hi].contiguous()


class _RS(torch.autograd.Function):
    """partial [N, D] -> this rank's block of the sum; backward: all-gather of the block gradients."""

    @staticmethod
    def forward(ctx, part, B):
        ctx.B = B
        return _reduce_scatter_blocks(part.contiguous(), B)

    @staticmethod
    def backward(ctx, g):
        return _gather_blocks(g.contiguous(), ctx.B), None


class _AG(torch.autograd.Function):
    """block -> full [N, D] table for gathers by arbitrary endpoint; backward: reduce-scatter of the (partial) table gradients."""

    @staticmethod
    def forward(ctx, block, B):
        ctx.B = B
        return _gather_blocks(block.contiguous(), B)

    @staticmethod
    def backward(ctx, g):
        return _reduce_scatter_blocks(g.contiguous(), ctx.B), None


class _Halo(torch.autograd.Function):
    """block [b, D] -> source rows [b + halo, D]: appends the first row of the next non-empty block; backward returns that row's
    gradient to its owner.  One [R, D] all-gather each way."""

    @staticmethod
    def forward(ctx, block, B):
        ctx.B = B
        if not _comm():
            return block
        D = block.shape[1]
        first = block[:1].contiguous() if block.shape[0] > 0 else torch.zeros(1, D, dtype=block.dtype, device=block.device)
        rows = [torch.empty_like(first) for _ in range(B.world)]
        dist.all_gather(rows, first)
        return torch.cat([block, rows[B.halo_owner]], dim=0) if B.halo else block

    @staticmethod
    def backward(ctx, g):
        B = ctx.B
        if not _comm():
            return g, None
        D = g.shape[1]
        gb = g[:B.rows].clone()
        mine = g[B.rows:B.rows + 1].contiguous() if B.halo else torch.zeros(1, D, dtype=g.dtype, device=g.device)
        rows = [torch.empty_like(mine) for _ in range(B.world)]
        dist.all_gather(rows, mine)
        for r in B.halo_users:                           # ranks that used my first row as their halo
            gb[0] += rows[r][0]
        return gb, None


class _EmbedRows(torch.autograd.Function):
    """rows [n, D] -> [N, D] with the rows at offset `a`, zeros elsewhere (the SpMM gathers by global source id)."""

    @staticmethod
    def forward(ctx, rows, a, N):
        ctx.a, ctx.n = a, rows.shape[0]
        full = torch.zeros(N, rows.shape[1], dtype=rows.dtype, device=rows.device)
        full[a:a + rows.shape[0]] = rows
        return full

    @staticmethod
    def backward(ctx, g):
        return g[ctx.a:ctx.a + ctx.n].contiguous(), None, None


class _BiasActRows(torch.autograd.Function):
    """Layer epilogue on a block of rows; dropout rows are global node ids (sgs_bias_act_rows)."""

    @staticmethod
    def forward(ctx, X, bias, row0, act, p, seed, site):
        L = _lib.lib()
        n, D = X.shape
        Y = torch.empty_like(X)
        if n > 0:
            _lib.check(L.sgs_bias_act_rows(ops._ptr(X.contiguous()), ops._ptr(bias), n, D, int(row0), act, float(p), seed, site, ops._ptr(Y),
                                           ops._stream()), "sgs_bias_act_rows")
        ctx.act, ctx.p, ctx.has_bias = act, float(p), bias is not None
        ctx.save_for_backward(Y if act != ops.ACT_NONE else None)
        return Y

    @staticmethod
    def backward(ctx, dY):
        L = _lib.lib()
        (Y,) = ctx.saved_tensors
        dY = dY.contiguous()
        if ctx.act != ops.ACT_NONE and dY.numel() > 0:
            dZ = torch.empty_like(dY)
            _lib.check(L.sgs_act_bwd(ops._ptr(dY), ops._ptr(Y), dY.numel(), ctx.act, ctx.p, ops._ptr(dZ), ops._stream()), "sgs_act_bwd")
        else:
            dZ = dY
        db = None
        if ctx.has_bias:
            db = ops._colsum(dZ) if dZ.shape[0] > 0 else torch.zeros(dZ.shape[1], dtype=dZ.dtype, device=dZ.device)
        return dZ, db, None, None, None, None, None


class _BlockCE(torch.autograd.Function):
    """This block's share of nn.CrossEntropyLoss()(logits[train], y[train]): sum of the block's train-row losses / #train rows of
    the WHOLE graph, so the ranks' values add up to the replicated loss (training_hybrid.py:113)."""

    @staticmethod
    def forward(ctx, logits_b, yb, mb_u8, n_train):
        L = _lib.lib()
        n, C = logits_b.shape
        dev = logits_b.device
        if n == 0:
            ctx.empty, ctx.C = True, C
            return torch.zeros((), dtype=torch.float32, device=dev)
        ctx.empty = False
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        row_lse, rowloss = torch.empty(n, dtype=torch.float32, device=dev), torch.empty(n, dtype=torch.float32, device=dev)
        nb = torch.empty(1, dtype=torch.int32, device=dev)
        _lib.check(L.sgs_masked_ce_fwd(ops._ptr(logits_b), n, C, ops._ptr(yb), ops._ptr(mb_u8), ops._ptr(loss), ops._ptr(row_lse), ops._ptr(rowloss),
                                       ops._ptr(nb), ops._stream()), "sgs_masked_ce_fwd")
        ctx.save_for_backward(logits_b, yb, mb_u8, row_lse, n_train)
        return rowloss.sum() / n_train[0].to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        if ctx.empty:                                    # an empty gradient, not None: the layers above hold collectives every rank must enter
            return torch.zeros(0, ctx.C, dtype=torch.float32, device=g.device), None, None, None
        L = _lib.lib()
        logits_b, yb, mb_u8, row_lse, n_train = ctx.saved_tensors
        n, C = logits_b.shape
        g = g.reshape(1).contiguous().float()
        d = torch.empty_like(logits_b)
        _lib.check(L.sgs_masked_ce_bwd(ops._ptr(logits_b), n, C, ops._ptr(yb), ops._ptr(mb_u8), ops._ptr(row_lse), ops._ptr(n_train), ops._ptr(g),
                                       ops._ptr(d), ops._stream()), "sgs_masked_ce_bwd")
        return d, None, None, None


class _ShardedNormBlock(torch.autograd.Function):
    """As _ShardedNorm, with the self-loop term owned (added and differentiated) by the rank that owns the node."""

    @staticmethod
    def forward(ctx, w, graph, box, lo, hi):
        nm = sharded_norm(graph, w)
        keep = torch.zeros_like(nm.what_loop)
        keep[lo:hi] = 1.0
        nm.what_loop = nm.what_loop * keep
        box.append(nm)
        ctx.nm, ctx.keep = nm, keep
        return torch.empty(graph.n_edges + graph.N, dtype=torch.float32, device=w.device)

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        nm, gr = ctx.nm, ctx.nm.graph
        g = g.contiguous()
        n, N = gr.n_edges, gr.N
        gw = g[:n].contiguous() if n > 0 else torch.zeros(1, dtype=torch.float32, device=g.device)
        gl = (g[n:] * ctx.keep).contiguous()
        Hn = torch.empty(N, dtype=torch.float32, device=g.device)
        wv = nm.w if n > 0 else torch.zeros(1, dtype=torch.float32, device=g.device)
        _lib.check(L.sgs_gcn_norm_bwd_node(ops._ptr(wv), ops._ptr(gw), ops._ptr(gl), n, N, ops._ptr(nm.dis), ops._ptr(nm.loopw),
                                           ops._ptr(gr.in_ptr), ops._ptr(gr.in_src), ops._ptr(gr.in_eid), ops._ptr(gr.out_ptr),
                                           ops._ptr(gr.out_dst), ops._ptr(gr.out_eid), ops._ptr(Hn), ops._stream()), "sgs_gcn_norm_bwd_node")
        if _comm():
            dist.all_reduce(Hn)
        dw = torch.empty(n, dtype=torch.float32, device=g.device)
        if n > 0:
            _lib.check(L.sgs_gcn_norm_bwd_edge(ops._ptr(gw), ops._ptr(gl), n, N, ops._ptr(nm.dis), ops._ptr(gr.loop_eid),
                                               ops._ptr(gr.edge_index), ops._ptr(Hn), ops._ptr(dw), ops._stream()), "sgs_gcn_norm_bwd_edge")
        return dw, None, None, None, None


def _block_norm(graph, w, B: NodeBlocks):
    if w is None or not (w.requires_grad and torch.is_grad_enabled()):
        nm = sharded_norm(graph, None if w is None else w.detach().contiguous())
        keep = torch.zeros_like(nm.what_loop)
        keep[B.lo:B.hi] = 1.0
        nm.what_loop = nm.what_loop * keep
        return nm
    box = []
    handle = _ShardedNormBlock.apply(w.contiguous(), graph, box, B.lo, B.hi)
    nm = box[0]
    nm.handle = handle
    return nm


def block_gcn_layer(src_rows, W, bias, nm, B: NodeBlocks, act=ops.ACT_NONE, p=0.0, seed=0, site=0):
    """One GCN layer in node-block form.  `src_rows` [rows + halo, Fin]: the layer input for this rank's source rows; returns the
    layer output for this rank's BLOCK [rows, D]."""
    xl_rows = ops.linear_nobias(src_rows, W)                                # 1 / R of the node-level GEMM; dW is a partial sum
    xl = _EmbedRows.apply(xl_rows, B.lo, B.N)
    part = ops._Propagate.apply(xl, nm.handle, None, nm, ops.ACT_NONE, 0.0, 0, 0)      # this rank's edges -> partial [N, D]
    yb = _RS.apply(part, B)
    return _BiasActRows.apply(yb, bias, B.lo, act, float(p), int(seed), int(site))


def train_step_blocksharded(args, model, shard: EdgeShard, optimizer_gnn, optimizer_edge_prob, criterion, q: int, noise=None):
    """train_step_sharded with the GCN layers in node-block form (reduce-scatter forward / all-gather backward, node-level work on
    1 / R of the rows): same draws (bit for bit), same logits and gradients up to fp32 summation order.  Returns the same trace dict;
    `learned_out` is the full [N, C] table (all-gathered for the regulariser anyway)."""
    from .model import _DropoutClock
    from .sampling import _NoiseClock
    noise = noise or {}
    model.train()
    optimizer_edge_prob.zero_grad()
    optimizer_gnn.zero_grad()
    sc = model.edge_prob_mlp
    x, N, ei = shard.x, shard.N, shard.edge_index
    p = sc.dropout.p
    act = ops.ACT_RELU_DROPOUT if p > 0 else ops.ACT_RELU
    off, bounds = shard.edge_offset, shard.bounds
    B = node_blocks(shard)
    x_rows = x[B.lo:B.hi + B.halo]
    yb, mb = shard.y[B.lo:B.hi].contiguous(), shard.train_mask[B.lo:B.hi].contiguous()

    seed_n, tick = (0, 0) if noise.get("prior") is not None else _NoiseClock.next()
    rs = dist_sample_topq(ops.SAMPLE_PRIOR, shard.prob, None, 0.0, q, ei, off, bounds, noise_local=noise.get("prior"), seed=seed_n,
                          stream_id=tick)
    g_r = ops.get_subgraph(ei, N, rs, eid=rs.eid - off)
    nm_r = _block_norm(g_r, None, B)
    h = block_gcn_layer(x_rows, sc.gcn1.lin.weight, sc.gcn1.bias, nm_r, B, act=act, p=p, seed=_DropoutClock.next_seed(), site=SITE_ENC)
    codes_b = block_gcn_layer(_Halo.apply(h, B), sc.gcn2.lin.weight, sc.gcn2.bias, nm_r, B, act=ops.ACT_RELU)
    codes = _AG.apply(codes_b, B)                                           # the scorer gathers codes by arbitrary endpoint
    active = ops.ActiveSet()
    p_local = ops.edge_score(codes, sc.fc1.weight, sc.fc1.bias, sc.fc2.weight, sc.fc2.bias, ei, active=active, p=p,
                             seed=_DropoutClock.next_seed(), site=SITE_SCORE, edge_id_offset=off)
    seed_n, tick = (0, 0) if noise.get("sample") is not None else _NoiseClock.next()
    smp = dist_sample_topq(ops.SAMPLE_LEARNED, p_local, shard.prob, args.degree_bias_coef, q, ei, off, bounds,
                           noise_local=noise.get("sample"), seed=seed_n, stream_id=tick)
    local_ids = smp.eid - off
    g_s = ops.get_subgraph(ei, N, smp, eid=local_ids)
    active.set(local_ids, g_s)
    w_local = p_local.index_select(0, local_ids)
    nm_s = _block_norm(g_s, w_local, B)
    pg = model.dropout.p
    actg = ops.ACT_RELU_DROPOUT if pg > 0 else ops.ACT_RELU
    h1 = block_gcn_layer(x_rows, model.gcn1.lin.weight, model.gcn1.bias, nm_s, B, act=actg, p=pg, seed=_DropoutClock.next_seed(), site=SITE_GNN)
    out_b = block_gcn_layer(_Halo.apply(h1, B), model.gcn2.lin.weight, model.gcn2.bias, nm_s, B)
    update_edge_mlp, random_b, counts = True, None, None
    dev = x.device
    if args.conditional:
        h1r = block_gcn_layer(x_rows, model.gcn1.lin.weight, model.gcn1.bias, nm_r, B, act=actg, p=pg, seed=_DropoutClock.next_seed(), site=SITE_GNN)
        random_b = block_gcn_layer(_Halo.apply(h1r, B), model.gcn2.lin.weight, model.gcn2.bias, nm_r, B)
        cbuf = torch.zeros(4, dtype=torch.int32, device=dev)
        if B.rows > 0:
            ops.masked_correct(out_b, yb, mb, out=cbuf[0:2])
            ops.masked_correct(random_b, yb, mb, out=cbuf[2:4])
        if _comm():
            dist.all_reduce(cbuf)                                           # gate counts: block sums -> global
        counts = cbuf.tolist()
        update_edge_mlp = counts[0] > counts[2]

    mb_u8 = ops._u8(mb)

    def block_ce(logits_b):                                                 # the ranks' values sum to the replicated cross entropy
        return _BlockCE.apply(logits_b.contiguous(), yb, mb_u8, B.n_train)

    learned_out, reg = None, None
    if update_edge_mlp:
        ce = block_ce(out_b)
        loss = ce
        c1 = args.regularizer1_coef if args.reg1 else 0.0
        c2 = args.consist_reg_coef if args.reg2 else 0.0
        learned_out = _AG.apply(out_b, B)                                   # the consistency regulariser gathers logits by endpoint
        if c1 != 0.0 or c2 != 0.0:
            reg = _ShardedEdgeReg.apply(w_local.contiguous(), learned_out.contiguous(), smp.edge_index, shard.y, ops._u8(shard.train_mask),
                                        g_s, float(c1), float(c2), q)
            loss = ce + reg                        # the global value on every rank; its backward yields this rank's edges' share only
        loss.backward()
    else:
        ce = block_ce(random_b)
        ce.backward()
    # every parameter gradient is a partial sum: ONE flat all-reduce
    if _comm():                                    # (every rank built the same autograd graph, so the same parameters hold gradients)
        params = [p_ for p_ in model.parameters() if p_.grad is not None]
        flat = torch.cat([p_.grad.reshape(-1) for p_ in params])
        dist.all_reduce(flat)
        o = 0
        for p_ in params:
            n_ = p_.numel()
            p_.grad.copy_(flat[o:o + n_].view_as(p_))
            o += n_
    total = ce.detach().clone()
    if _comm():
        dist.all_reduce(total)
    if reg is not None:
        total = total + reg.detach()
    if update_edge_mlp:
        optimizer_edge_prob.step()
    optimizer_gnn.step()
    if learned_out is None:
        learned_out = _gather_blocks(out_b.detach(), B)
    return dict(loss=total, sample=smp, random=rs, update_edge_mlp=update_edge_mlp, learned_out=learned_out.detach(), counts=counts)
