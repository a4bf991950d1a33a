"""Host-side wrappers over the C ABI (include/sgs_hip.h): tensor checks, workspace, stream
plumbing and the torch.autograd.Function glue.  PyTorch is used for device memory, streams
and autograd bookkeeping only; every numeric step of the hot path runs in libsgs_hip.so.
Nothing here computes on the CPU: a non-HIP tensor raises."""
from __future__ import annotations

import os

import numpy as np
import torch

from . import _lib

SAMPLE_LEARNED, SAMPLE_PRIOR = 0, 1
ACT_NONE, ACT_RELU, ACT_RELU_DROPOUT = 0, 1, 2

_workspaces = {}
_retired_workspaces = []      # arenas outgrown while HIP graphs may still reference them (pin_workspaces)
_pin_workspaces = False


def _need_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("sgs_gnn_amd: the SGS hot path runs only on a HIP device (got a CPU tensor); "
                               "there is no CPU fallback")


def _ptr(t, dtype=None):
    if t is None:
        return None
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f"sgs_gnn_amd: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError("sgs_gnn_amd: tensor must be contiguous")
    return t.data_ptr()


try:
    _raw_stream = torch._C._cuda_getCurrentRawStream            # fast path: no Stream object per call
except AttributeError:                                         # pragma: no cover
    _raw_stream = None


def _stream():
    """hipStream_t of torch's current stream on the current device, as an integer."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


_ws_slot = 0


class workspace_slot:
    """Kernels launched inside this context take their scratch from arena `slot` instead of arena 0.  An arena is reused in
    stream order, so work that may run CONCURRENTLY with the main stream's (stepgraph.py replays the parameter-independent
    prefix of the next partition on a second stream) must be recorded with its own."""

    def __init__(self, slot: int):
        self.slot, self.prev = int(slot), 0

    def __enter__(self):
        global _ws_slot
        self.prev, _ws_slot = _ws_slot, self.slot
        return self

    def __exit__(self, *exc):
        global _ws_slot
        _ws_slot = self.prev
        return False


# Stream ownership of the arenas.  An arena is reused in STREAM ORDER: two kernels that take scratch from the same slot are safe only
# if one stream orders them.  Round 2 lost a bench run to exactly this (a partition's CSR build issued on the prefetch stream took
# scratch from arena 0 while the replayed step's sampler was using it) and fixed it by convention.  The
# guard makes the convention checkable: each slot remembers the raw stream that last took scratch from it, and -- with
# SGS_WS_GUARD=1 (the test suite sets it) -- a request from ANOTHER stream raises, unless the caller declared the hand-over
# (`workspace_handover`: "the current stream has been ordered after the owner", e.g. right after stream.wait_stream(owner)).
_ws_owner = {}
_ws_guard = os.environ.get("SGS_WS_GUARD") == "1"


def set_workspace_guard(on: bool) -> None:
    global _ws_guard
    _ws_guard = bool(on)
    _ws_owner.clear()


def workspace_handover(device=None, slot=None) -> None:
    """The caller has made the CURRENT stream wait for everything issued so far on this device's arenas' owners (wait_stream /
    wait_event / a device synchronisation): the current stream owns every slot (or just `slot`) from here on."""
    if not _ws_guard:
        return
    dev = torch.cuda.current_device() if device is None else (device.index if device.index is not None else torch.cuda.current_device())
    cur = _stream()
    if slot is not None:
        _ws_owner[(dev, int(slot))] = cur
        return
    for key in list(_ws_owner):
        if key[0] == dev:
            _ws_owner[key] = cur


def workspace(nbytes: int, device) -> torch.Tensor:
    """Grow-only per-device scratch arena (stream-ordered reuse on the current stream; see workspace_slot)."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), _ws_slot)
    if _ws_guard:
        cur = _stream()
        own = _ws_owner.setdefault(key, cur)
        if own != cur:
            raise RuntimeError(f"sgs_gnn_amd: scratch arena {key[1]} of device {key[0]} is owned by stream {own:#x} and was asked for from stream "
                               f"{cur:#x} without a hand-over (ops.workspace_handover): kernels on two streams would share scratch memory; "
                               "use ops.workspace_slot(k) for work that runs beside the main stream")
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        if ws is not None and _pin_workspaces:
            _retired_workspaces.append(ws)
        ws = torch.empty(max(int(nbytes * 1.25), 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def pin_workspaces(on: bool = True) -> None:
    """Once captured HIP graphs hold the arena's address, an outgrown arena must stay allocated."""
    global _pin_workspaces
    _pin_workspaces = _pin_workspaces or bool(on)


# ------------------------------------------------------------------ memo scope
_memo_scope = 0


def memo_scope() -> int:
    """Identifier of the current memo scope: per-step memos (GCNConv's x W^T shared by the learned and the random forward of one
    step) are keyed on it, so they can only be hit inside the step that made them."""
    return _memo_scope


def new_memo_scope() -> int:
    """Open a new scope (called once per batch by train / evaluate): every memo made before is dead from here on."""
    global _memo_scope
    _memo_scope += 1
    return _memo_scope


def drop_memos(module) -> None:
    """Forget every memoised x W^T held by `module`'s layers (after a capture: the memo would point into a graph's pool)."""
    for mod in module.modules():
        if getattr(mod, "_lin_cache", None) is not None:
            mod._lin_cache = None


# ------------------------------------------------------------------ randomness
_rng_epoch = None       # keeps the registered device word alive


def set_rng_epoch_buffer(epoch) -> None:
    """Register (or clear with None) the device word every RNG-consuming kernel folds into its seed
    (sgs_rng_set_epoch_buffer): a HIP graph that increments it draws fresh noise on every replay."""
    global _rng_epoch
    L = _lib.lib()
    if epoch is not None:
        _need_gpu(epoch)
        if epoch.dtype != torch.int64 or epoch.numel() != 1:
            raise RuntimeError("sgs_gnn_amd: the RNG epoch buffer is one int64 device word")
    _lib.check(L.sgs_rng_set_epoch_buffer(None if epoch is None else epoch.data_ptr()), "sgs_rng_set_epoch_buffer")
    _rng_epoch = epoch


_dyn_edges = None       # keeps the registered device word alive


def set_dyn_edges(word) -> None:
    """Register (or clear with None) the device word holding the live candidate-edge count (sgs_dyn_edges_set): kernels over the
    candidate edges launched while it is registered -- i.e. captured into a HIP graph -- take their size from it at run time."""
    global _dyn_edges
    L = _lib.lib()
    if word is not None:
        _need_gpu(word)
        if word.dtype != torch.int64 or word.numel() not in (1, 2):
            raise RuntimeError("sgs_gnn_amd: the dynamic sizes are one or two int64 device words (live E[, live canonical edges])")
    _lib.check(L.sgs_dyn_edges_set(None if word is None else word.data_ptr()), "sgs_dyn_edges_set")
    _dyn_edges = word


def exp_noise(seed: int, stream_id: int, E: int, device) -> torch.Tensor:
    L = _lib.lib()
    out = torch.empty(E, dtype=torch.float32, device=device)
    _need_gpu(out)
    _lib.check(L.sgs_exp_noise(seed, stream_id, E, _ptr(out), _stream()), "sgs_exp_noise")
    return out


def dropout_keep(seed: int, site: int, rows: int, cols: int, p: float, device) -> torch.Tensor:
    L = _lib.lib()
    out = torch.empty(rows, cols, dtype=torch.uint8, device=device)
    _need_gpu(out)
    _lib.check(L.sgs_dropout_keep(seed, site, rows, cols, p, _ptr(out), _stream()), "sgs_dropout_keep")
    return out.bool()


# ------------------------------------------------------------------ sampler
class SampleResult:
    __slots__ = ("mask", "eid", "edge_index", "p", "stats", "keys", "E", "q")

    def check(self) -> None:
        """torch.multinomial(replacement=False) raises when fewer than q categories have a positive weight; the fused draw cannot
        raise from the device.  It reports the case through `stats`: the threshold key is then 0 (zero-weight edges were admitted
        by the lowest-id tie-break).  Calling this reads stats back (one synchronisation) and raises like the reference."""
        if self.q > 0 and float(self.stats[2]) <= 0.0:
            raise RuntimeError("invalid multinomial distribution (with replacement=False, not enough non-negative category to sample)")


def sample_topq(mode: int, p: torch.Tensor, prior, c: float, q: int, edge_index, noise=None, seed: int = 0,
                stream_id: int = 0, want_keys: bool = False, want_p: bool = True) -> SampleResult:
    """K0/K2/K3 (see sgs_sample_topq).  p [E] f32 (None: uniform weights); prior [E] f32 or None; edge_index [2,E] i64."""
    L = _lib.lib()
    _need_gpu(p, prior, edge_index, noise)
    if p is None and edge_index is None:
        raise RuntimeError("sample_topq: uniform weights (p=None) need edge_index for the number of candidates")
    E = p.numel() if p is not None else edge_index.shape[1]
    dev = p.device if p is not None else edge_index.device
    if q > E:
        raise RuntimeError(f"cannot sample q={q} > E={E} edges without replacement")
    r = SampleResult()
    r.E, r.q = E, q
    r.mask = torch.empty(E, dtype=torch.bool, device=dev)
    r.eid = torch.empty(q, dtype=torch.int64, device=dev)
    r.edge_index = torch.empty(2, q, dtype=torch.int64, device=dev) if edge_index is not None else None
    r.p = torch.empty(q, dtype=torch.float32, device=dev) if (want_p and p is not None) else None
    r.stats = torch.empty(4, dtype=torch.float32, device=dev)
    r.keys = torch.empty(E, dtype=torch.float32, device=dev) if want_keys else None
    nws = L.sgs_sample_topq_workspace_bytes(E)
    ws = workspace(nws, dev)
    _lib.check(L.sgs_sample_topq(mode, _ptr(p, torch.float32), _ptr(prior, torch.float32), float(c),
                                 _ptr(noise, torch.float32), seed, stream_id, E, q, _ptr(edge_index, torch.int64),
                                 _ptr(r.mask), _ptr(r.eid), _ptr(r.edge_index), _ptr(r.p), _ptr(r.stats),
                                 _ptr(r.keys), ws.data_ptr(), ws.numel(), _stream()), "sgs_sample_topq")
    return r


def gather_columns(edge_index: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """edge_index[:, idx] (sampling.py:163) as one launch."""
    L = _lib.lib()
    _need_gpu(edge_index, idx)
    q = idx.numel()
    out = torch.empty(2, q, dtype=torch.int64, device=edge_index.device)
    _lib.check(L.sgs_gather_columns(_ptr(edge_index, torch.int64), edge_index.shape[1], _ptr(idx.contiguous(), torch.int64), q, _ptr(out),
                                    _stream()), "sgs_gather_columns")
    return out


class _STWeights(torch.autograd.Function):
    """sampling.py:137-138,155: clamp(p * ((one_hot - s).detach() + s), 0, 1)[mask]."""

    @staticmethod
    def forward(ctx, p, prior, c, stats, eid):
        L = _lib.lib()
        E, q = p.numel(), eid.numel()
        w = torch.empty(q, dtype=torch.float32, device=p.device)
        _lib.check(L.sgs_st_weights_fwd(_ptr(p, torch.float32), _ptr(prior, torch.float32), float(c), _ptr(stats),
                                        _ptr(eid, torch.int64), E, q, _ptr(w), _stream()), "sgs_st_weights_fwd")
        ctx.save_for_backward(p, prior if prior is not None else torch.empty(0, device=p.device), stats, eid)
        ctx.c, ctx.has_prior = float(c), prior is not None
        return w

    @staticmethod
    def backward(ctx, gw):
        L = _lib.lib()
        p, prior, stats, eid = ctx.saved_tensors
        prior = prior if ctx.has_prior else None
        E, q = p.numel(), eid.numel()
        gw = gw.contiguous()
        gp = torch.empty(E, dtype=torch.float32, device=p.device)
        nws = L.sgs_st_weights_bwd_workspace_bytes(E, q)
        ws = workspace(nws, p.device)
        _lib.check(L.sgs_st_weights_bwd(_ptr(p), _ptr(prior), ctx.c, _ptr(stats), _ptr(eid), _ptr(gw), E, q, _ptr(gp),
                                        ws.data_ptr(), ws.numel(), _stream()), "sgs_st_weights_bwd")
        return gp, None, None, None, None


_zero_tokens = {}


def _zero_token(device) -> torch.Tensor:
    """One persistent zero per device; `.expand(E)` of it is the "all zeros, look at ActiveSet.gq" gradient below (no launch)."""
    key = (device.type, device.index)
    t = _zero_tokens.get(key)
    if t is None:
        t = _zero_tokens[key] = torch.zeros(1, dtype=torch.float32, device=device)
    return t


class _SelectSampled(torch.autograd.Function):
    """probs[eid] with the VALUES the sampler's compaction already gathered (sgs_sample_topq's sampled_p): the forward
    launches nothing; the backward is index_select's (scatter of the q gradients into zeros [E]; the eids are unique).
    With `active` (the scorer's ActiveSet for these very edges, hybrid pipeline) the q gradients are handed to the scorer's
    backward directly (`active.gq`) and the [E] gradient that autograd wants is a stride-0 view of a persistent zero: the
    zero-fill of [E], the scatter and the scorer's gather back to [q] -- three launches -- disappear."""

    @staticmethod
    def forward(ctx, probs, eid, values, active):
        ctx.save_for_backward(eid)
        ctx.E, ctx.active = probs.numel(), active
        return values.view_as(values)

    @staticmethod
    def backward(ctx, g):
        (eid,) = ctx.saved_tensors
        act = ctx.active
        if act is not None and act.eid is not None and act.eid.data_ptr() == eid.data_ptr() and act.eid.numel() == eid.numel():
            act.gq = g.contiguous()
            return _zero_token(g.device).expand(ctx.E), None, None, None
        gp = torch.zeros(ctx.E, dtype=g.dtype, device=g.device)
        gp.index_copy_(0, eid, g.contiguous())
        return gp, None, None, None


def select_sampled(probs, eid, values, active=None):
    _need_gpu(probs, eid, values)
    return _SelectSampled.apply(probs, eid, values, active)


def st_weights(p, prior, c, stats, eid):
    _need_gpu(p, prior, stats, eid)
    return _STWeights.apply(p.contiguous(), prior, c, stats, eid)


# ------------------------------------------------------------------ graph + GCN
class Graph:
    """Both CSR orientations of one edge list (sgs_graph_build).  Built once per sampled graph
    and shared by every layer / pass that runs over it."""

    def __init__(self, edge_index: torch.Tensor, N: int):
        L = _lib.lib()
        _need_gpu(edge_index)
        if edge_index.dtype != torch.int64 or edge_index.dim() != 2 or edge_index.shape[0] != 2:
            raise RuntimeError("edge_index must be an int64 [2, E] tensor")
        ei = edge_index.contiguous()
        dev = ei.device
        n = ei.shape[1]
        self.edge_index, self.n_edges, self.N = ei, n, int(N)
        # one allocation, seven views (host time matters: the step is launch-bound at partition scale)
        ne = max(n, 1)
        sizes = [N + 1, N + 1, ne, ne, ne, ne, max(N, 1)]
        offs = [0]
        for z in sizes:
            offs.append(offs[-1] + ((z + 63) & ~63))             # keep every view 256-B aligned
        buf = torch.empty(offs[-1], dtype=torch.int32, device=dev)
        (self.in_ptr, self.out_ptr, self.in_src, self.in_eid, self.out_dst, self.out_eid, self.loop_eid) = (
            buf[offs[i]:offs[i] + sizes[i]] for i in range(7))
        nws = L.sgs_graph_build_workspace_bytes(n, N)
        ws = workspace(nws, dev)
        _lib.check(L.sgs_graph_build(_ptr(ei), n, N, _ptr(self.in_ptr), _ptr(self.in_src), _ptr(self.in_eid),
                                     _ptr(self.out_ptr), _ptr(self.out_dst), _ptr(self.out_eid), _ptr(self.loop_eid),
                                     ws.data_ptr(), ws.numel(), _stream()), "sgs_graph_build")


_SORT_SUBGRAPH_EDGES = int(os.environ.get("SGS_SORT_SUBGRAPH_EDGES", 1 << 22))     # drawn edges from which get_subgraph sorts instead of filtering


def get_subgraph(parent_edge_index: torch.Tensor, N: int, sample, eid=None) -> Graph:
    """CSR of a drawn subgraph (`sample` = SampleResult of a draw over `parent_edge_index`) squeezed out of the parent's cached
    CSR (sgs_graph_filter): no atomics and no per-row sort, identical arrays to Graph(sample.edge_index).  The result is cached
    on `sample.edge_index`, so every later get_graph() on the drawn edge list reuses it.  `eid`: the selected edges' positions in
    `parent_edge_index` when they differ from `sample.eid` (edge-sharded draws report GLOBAL ids: pass the local ones)."""
    L = _lib.lib()
    parent = get_graph(parent_edge_index, N)
    ei = sample.edge_index
    n = ei.shape[1]
    g = Graph.__new__(Graph)
    g.edge_index, g.n_edges, g.N = ei, n, int(N)
    ne = max(n, 1)
    sizes = [N + 1, N + 1, ne, ne, ne, ne, max(N, 1)]
    offs = [0]
    for z in sizes:
        offs.append(offs[-1] + ((z + 63) & ~63))
    buf = torch.empty(offs[-1], dtype=torch.int32, device=ei.device)
    (g.in_ptr, g.out_ptr, g.in_src, g.in_eid, g.out_dst, g.out_eid, g.loop_eid) = (buf[offs[i]:offs[i] + sizes[i]] for i in range(7))
    if n >= _SORT_SUBGRAPH_EDGES and src_sorted(parent_edge_index):
        # whole-graph scale: one packed radix sort of the DRAWN edges instead of two passes over the parent's CSR (sgs_graph_build_src_sorted)
        ws = workspace(L.sgs_graph_build_src_sorted_workspace_bytes(n, N), ei.device)
        _lib.check(L.sgs_graph_build_src_sorted(_ptr(ei), n, N, _ptr(g.in_ptr), _ptr(g.in_src), _ptr(g.in_eid), _ptr(g.out_ptr), _ptr(g.out_dst),
                                                _ptr(g.out_eid), _ptr(g.loop_eid), None, ws.data_ptr(), ws.numel(), _stream()),
                   "sgs_graph_build_src_sorted")
        try:
            ei._sgs_graph = g
            ei._sgs_graph_version = ei._version
        except Exception:
            pass
        return g
    ws = workspace(L.sgs_graph_filter_workspace_bytes(parent.n_edges, N), ei.device)
    _lib.check(L.sgs_graph_filter(_ptr(parent.in_ptr), _ptr(parent.in_src), _ptr(parent.in_eid), _ptr(parent.out_ptr), _ptr(parent.out_dst),
                                  _ptr(parent.out_eid), parent.n_edges, N, _ptr(_u8(sample.mask)), _ptr(sample.eid if eid is None else eid, torch.int64), n,
                                  _ptr(g.in_ptr), _ptr(g.in_src), _ptr(g.in_eid), _ptr(g.out_ptr), _ptr(g.out_dst), _ptr(g.out_eid),
                                  _ptr(g.loop_eid), ws.data_ptr(), ws.numel(), _stream()), "sgs_graph_filter")
    try:
        ei._sgs_graph = g
        ei._sgs_graph_version = ei._version
    except Exception:
        pass
    return g


def get_graph(edge_index: torch.Tensor, N: int) -> Graph:
    """Graph for `edge_index`, cached ON the tensor object (dies with it; keyed by its version
    counter), so the encoder and the GNN that receive the same tensor share one build."""
    g = getattr(edge_index, "_sgs_graph", None)
    if g is None or g.N != N or getattr(edge_index, "_sgs_graph_version", -1) != edge_index._version:
        g = Graph(edge_index, N)
        try:
            edge_index._sgs_graph = g
            edge_index._sgs_graph_version = edge_index._version
        except Exception:
            pass
    return g


def get_pairs(edge_index: torch.Tensor, N: int, build: bool = False):
    """(canon int32 [M], mate int32 [E]) of the paired scorer forward (sgs_edge_mates / sgs_edge_score_fwd_paired), cached on the
    tensor like its Graph; None when it has not been built.  `build=True` builds it (two launches over the cached CSR + one
    compaction with a read-back of M: set-up work, done once per partition by the trainers, never inside a captured step)."""
    c = getattr(edge_index, "_sgs_pairs", None)
    if c is not None and c[2] == edge_index._version:
        return c[0], c[1]
    if not build:
        return None
    L = _lib.lib()
    _need_gpu(edge_index)
    g = get_graph(edge_index, N)
    E = g.n_edges
    dev = edge_index.device
    mate = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
    if E > 0:
        ws = workspace(L.sgs_edge_mates_workspace_bytes(E), dev)
        _lib.check(L.sgs_edge_mates(_ptr(g.edge_index), E, N, _ptr(g.out_ptr), _ptr(g.out_dst), _ptr(g.out_eid), _ptr(mate), ws.data_ptr(),
                                    ws.numel(), _stream()), "sgs_edge_mates")
    ar = torch.arange(E, dtype=torch.int32, device=dev)
    canon = ar[(mate[:E] < 0) | (ar < mate[:E])].contiguous()
    try:
        edge_index._sgs_pairs = (canon, mate, edge_index._version)
    except Exception:
        pass
    return canon, mate


class Norm:
    """gcn_norm result for (graph, w): dis, loopw and the normalised weights in both CSR orders.
    `handle` is the autograd edge through which the layers' gradients wrt the normalised
    weights ([n_edges] edge order + [N] loops) flow back to `w`; its storage is never read."""
    __slots__ = ("graph", "w", "dis", "loopw", "what_in", "what_out", "what_loop", "handle", "_g_first", "_g_extra", "_dw_first", "_park_ok", "__weakref__")


def _norm_forward(graph: Graph, w):
    L = _lib.lib()
    dev = graph.edge_index.device
    nm = Norm()
    nm.graph, nm.w = graph, w
    ne, Nn = max(graph.n_edges, 1), graph.N
    sizes = [Nn, Nn, ne, ne, Nn]
    offs = [0]
    for z in sizes:
        offs.append(offs[-1] + ((z + 63) & ~63))
    buf = torch.empty(offs[-1], dtype=torch.float32, device=dev)
    nm.dis, nm.loopw, nm.what_in, nm.what_out, nm.what_loop = (buf[offs[i]:offs[i] + sizes[i]] for i in range(5))
    nm.handle = None
    _lib.check(L.sgs_gcn_norm_fwd(_ptr(w, torch.float32), graph.n_edges, graph.N, _ptr(graph.in_ptr), _ptr(graph.in_src),
                                  _ptr(graph.in_eid), _ptr(graph.out_ptr), _ptr(graph.out_dst), _ptr(graph.out_eid),
                                  _ptr(graph.loop_eid), _ptr(nm.dis), _ptr(nm.loopw), _ptr(nm.what_in), _ptr(nm.what_out),
                                  _ptr(nm.what_loop), _stream()), "sgs_gcn_norm_fwd")
    return nm


class _GCNNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w, graph, box):
        nm = _norm_forward(graph, w)
        box.append(nm)
        ctx.nm = nm
        return torch.empty(graph.n_edges + graph.N, dtype=torch.float32, device=w.device)

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        nm, gr = ctx.nm, ctx.nm.graph
        g = g.contiguous()
        n = gr.n_edges
        # side bands (see _handle_grad / note_first_dw): a second layer's gradient wrt the normalised weights is summed on read, and the
        # edge weights' other consumer's d w (the loss's regularisers) is accumulated IN PLACE -- two autograd add launches less
        extra = getattr(nm, "_g_extra", None)
        nm._g_first = nm._g_extra = None
        first = getattr(nm, "_dw_first", None)
        first = first() if first is not None else None
        nm._dw_first = None
        if first is not None and (first.numel() != n or not first.is_contiguous() or first.dtype != torch.float32):
            first = None
        dw = first if first is not None else torch.empty(n, dtype=torch.float32, device=g.device)
        if n > 0:
            gw, gl = g[:n], g[n:]
            g2w = g2l = None
            if extra is not None and extra.numel() == g.numel():
                g2w, g2l = extra[:n].data_ptr(), extra[n:].data_ptr()
            elif extra is not None:
                g = g + extra
                gw, gl = g[:n], g[n:]
            nws = L.sgs_gcn_norm_bwd_workspace_bytes(gr.N)
            ws = workspace(nws, g.device)
            _lib.check(L.sgs_gcn_norm_bwd_sum(_ptr(nm.w), gw.data_ptr(), gl.data_ptr(), g2w, g2l, None if first is None else first.data_ptr(), n, gr.N,
                                              _ptr(nm.dis), _ptr(nm.loopw), _ptr(gr.in_ptr), _ptr(gr.in_src), _ptr(gr.in_eid), _ptr(gr.out_ptr),
                                              _ptr(gr.out_dst), _ptr(gr.out_eid), _ptr(gr.loop_eid), _ptr(gr.edge_index), _ptr(dw), ws.data_ptr(),
                                              ws.numel(), _stream()), "sgs_gcn_norm_bwd_sum")
        return (None if first is not None else dw), None, None


def _handle_grad(nm, g):
    """What a layer's backward returns for the normalisation's handle.  Both layers of a model share one Norm; autograd would add their two
    gradients with a launch of its own before _GCNNorm.backward.  Instead the first layer to finish returns its gradient as usual and the
    second parks its own on the Norm (summed on read by sgs_gcn_norm_bwd_sum) and returns None.  Nothing can get lost: the parked gradient
    has exactly one consumer, _GCNNorm.backward, which runs after every layer over the Norm has reported."""
    if not getattr(nm, "_park_ok", False):             # only Norms whose backward (_GCNNorm) reads the side band
        return g
    if getattr(nm, "_g_first", None) is None:
        nm._g_first = g
        return g
    if getattr(nm, "_g_extra", None) is None:
        nm._g_extra = g
        return None
    return g


def gcn_norm(graph: Graph, w=None) -> Norm:
    """K4.  `w` [n_edges] f32 or None (unit weights).  The unit-weight result depends on the graph alone and is kept on it:
    the scorer's encoder and the GNN's random forward normalise the same random subgraph (training_hybrid.py:52,93)."""
    if w is None:
        nm = getattr(graph, "_norm_unit", None)
        if nm is None:
            nm = graph._norm_unit = _norm_forward(graph, None)
        return nm
    _need_gpu(w)
    w = w.contiguous()
    if w.dtype != torch.float32 or w.numel() != graph.n_edges:
        raise RuntimeError(f"edge_weight must be float32 [{graph.n_edges}]")
    if not (w.requires_grad and torch.is_grad_enabled()):
        return _norm_forward(graph, w.detach())
    box = []
    handle = _GCNNorm.apply(w, graph, box)
    nm = box[0]
    nm.handle = handle
    nm._park_ok = True
    try:
        import weakref
        w._sgs_norm = weakref.ref(nm)          # (note_first_dw: the loss finds the normalisation that differentiates these weights)
    except Exception:
        pass
    return nm


def _spmm(X, ptr, col, val, diag, bias, act, p, seed, site, N, D, nnz):
    L = _lib.lib()
    Y = torch.empty(N, D, dtype=torch.float32, device=X.device)
    _lib.check(L.sgs_spmm_csr(_ptr(X, torch.float32), N, D, nnz, _ptr(ptr), _ptr(col), _ptr(val), _ptr(diag), _ptr(bias),
                              act, float(p), seed, site, _ptr(Y), _stream()), "sgs_spmm_csr")
    return Y


class _Propagate(torch.autograd.Function):
    """Y = act(A_hat X + bias); A_hat from `nm` (K5 forward + its three backward products)."""

    @staticmethod
    def forward(ctx, X, handle, bias, nm, act, p, seed, site):
        gr = nm.graph
        N, D = X.shape
        Y = _spmm(X, gr.in_ptr, gr.in_src, nm.what_in, nm.what_loop, bias, act, p, seed, site, N, D, gr.n_edges)
        ctx.nm, ctx.act, ctx.p = nm, act, p
        ctx.has_bias, ctx.has_handle = bias is not None, handle is not None
        ctx.save_for_backward(X, Y if act != ACT_NONE else None)
        return Y

    @staticmethod
    def backward(ctx, dY):
        L = _lib.lib()
        nm, gr = ctx.nm, ctx.nm.graph
        X, Y = ctx.saved_tensors
        N, D = X.shape
        dY = dY.contiguous()
        dX = dbias = g = None
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.act != ACT_NONE and want_db:
            dZ, dbias = _act_bwd_colsum(dY, Y, ctx.act, ctx.p)
        elif ctx.act != ACT_NONE:
            dZ = torch.empty_like(dY)
            _lib.check(L.sgs_act_bwd(_ptr(dY), _ptr(Y), dY.numel(), ctx.act, float(ctx.p), _ptr(dZ), _stream()), "sgs_act_bwd")
        else:
            dZ = dY
        if ctx.needs_input_grad[0]:
            dX = _spmm(dZ, gr.out_ptr, gr.out_dst, nm.what_out, nm.what_loop, None, ACT_NONE, 0.0, 0, 0, N, D, gr.n_edges)
        if ctx.has_handle and ctx.needs_input_grad[1]:
            g = torch.empty(gr.n_edges + gr.N, dtype=torch.float32, device=dY.device)
            gw, gl = g[:gr.n_edges], g[gr.n_edges:]
            _lib.check(L.sgs_sddmm_csr(_ptr(dZ), _ptr(X), N, D, gr.n_edges, _ptr(gr.in_ptr), _ptr(gr.in_src), _ptr(gr.in_eid),
                                       gw.data_ptr(), gl.data_ptr(), _stream()), "sgs_sddmm_csr")
        if want_db and dbias is None:
            dbias = _colsum(dZ)
        return dX, (_handle_grad(nm, g) if g is not None else None), dbias, None, None, None, None, None


def gcn_propagate(X, nm: Norm, bias=None, act=ACT_NONE, p=0.0, seed=0, site=0):
    """K5: act(A_hat X + bias) with autograd to X, bias and (through nm.handle) the edge weights."""
    _need_gpu(X, bias)
    if X.dtype != torch.float32 or X.dim() != 2 or X.shape[0] != nm.graph.N:
        raise RuntimeError("gcn_propagate: X must be float32 [N, D]")    # nm.what_loop may be None (no self-loop term)
    return _Propagate.apply(X.contiguous(), nm.handle, bias, nm, act, float(p), int(seed), int(site))


# ------------------------------------------------------------------ edge scorer (K1b)
class ActiveSet:
    """Which edges can carry a non-zero upstream gradient into the scorer's backward.
    None = all (dense backward over every scored edge).  The hybrid pipeline sets it to the
    q sampled edges after the draw (every other entry of dL/dp is exactly zero there,
    training_hybrid.py:86), which cuts the scorer's backward from E to q rows."""
    __slots__ = ("eid", "graph", "gq")

    def __init__(self):
        self.eid, self.graph, self.gq = None, None, None      # gq: the active rows' upstream gradient, handed over by _SelectSampled

    def set(self, eid: torch.Tensor, graph: Graph):
        self.eid, self.graph = eid, graph


def _act_bwd_colsum(dY, Y, act, p):
    """(dZ, colsum(dZ)) with dZ = dY * act'(Y): the activation and bias gradients of a layer in one pass."""
    L = _lib.lib()
    N, D = dY.shape
    dZ = torch.empty_like(dY)
    out = torch.empty(D, dtype=torch.float32, device=dY.device)
    ws = workspace(L.sgs_colsum_workspace_bytes(N, D), dY.device)
    _lib.check(L.sgs_act_bwd_colsum(_ptr(dY), _ptr(Y), N, D, act, float(p), _ptr(dZ), _ptr(out), ws.data_ptr(), ws.numel(), _stream()),
               "sgs_act_bwd_colsum")
    return dZ, out


def _colsum(A):
    L = _lib.lib()
    N, D = A.shape
    out = torch.empty(D, dtype=torch.float32, device=A.device)
    ws = workspace(L.sgs_colsum_workspace_bytes(N, D), A.device)
    _lib.check(L.sgs_colsum(_ptr(A), N, D, _ptr(out), ws.data_ptr(), ws.numel(), _stream()), "sgs_colsum")
    return out


def _endpoint_reduce(M_out, M_in, T, graph: Graph, s_out, s_in, H):
    L = _lib.lib()
    out = torch.empty(graph.N, H, dtype=torch.float32, device=M_out.device)
    _lib.check(L.sgs_endpoint_reduce(_ptr(M_out), _ptr(M_in), _ptr(T), graph.N, H, graph.n_edges, _ptr(graph.in_ptr), _ptr(graph.in_src), _ptr(graph.in_eid),
                                     _ptr(graph.out_ptr), _ptr(graph.out_dst), _ptr(graph.out_eid), float(s_out), float(s_in),
                                     _ptr(out), _stream()), "sgs_endpoint_reduce")
    return out


_mask_backward = True        # False: the dense fp32 dv path at every size (tests compare the two)
_fwd_mask = True             # False: the forward keeps no mask; the backward recomputes the hidden layer (sgs_edge_score_bwd_core_bits)
_fused_backward = os.environ.get("SGS_FUSED_BWD", "1") != "0"   # False: feat / dfeat as [n, H] arrays and sgs_endpoint_reduce_pair_bits (the form for edge lists not sorted by source)


def src_sorted(edge_index: torch.Tensor) -> bool:
    """Is the edge list sorted by source (PyG's coalesced / row-sorted layout: every loader of the reference emits it)?  Cached on the
    tensor like its Graph; computing it reads one word back, so it is never computed inside a stream capture (unknown = False there:
    the unfused backward is correct for any order).  stepgraph.py stamps its static slot tensors after checking the partitions."""
    c = getattr(edge_index, "_sgs_src_sorted", None)
    if c is not None and c[1] == edge_index._version:
        return c[0]
    if torch.cuda.is_current_stream_capturing():
        return False
    n = edge_index.shape[1]
    ok = True if n < 2 else bool((edge_index[0, 1:] >= edge_index[0, :-1]).all())
    try:
        edge_index._sgs_src_sorted = (ok, edge_index._version)
    except Exception:
        pass
    return ok


class _EdgeScore(torch.autograd.Function):
    """K1b with the node-level half of fc1 inside: U = codes W1b^T (library GEMM) in forward; in backward d codes gets dU W1b on
    top of the direct term, and BOTH halves of d fc1.weight [H, 2H] are written in place by the two weight-gradient GEMMs
    (d W1a = dv^T feat, d W1b = dU^T codes; sgs_gemm_tn_ld with ldc = 2H) -- no slice views, zero fills or gradient adds."""

    @staticmethod
    def forward(ctx, codes, W1, b1, w2, b2, edge_index, active, p, seed, site, edge_id_offset, pairs):
        L = _lib.lib()
        N, H = codes.shape
        E = edge_index.shape[1]
        U = torch.mm(codes, W1[:, H:].t())
        out = torch.empty(E, dtype=torch.float32, device=codes.device)
        ws = workspace(L.sgs_edge_score_workspace_bytes(N, H, E), codes.device)
        maskbits = None
        if (_fwd_mask and _mask_backward and E >= 65536 and L.sgs_edge_score_bwd_bits_supported(H) and ctx.needs_input_grad[0]
                and _variant_overrides_are_default()):
            # a forward whose backward will follow: keep the ReLU x dropout mask of every scored edge (one bit per hidden unit), so that the
            # backward needs no recompute of the hidden layer (_edge_score_backward_mask)
            maskbits = torch.empty(E, H // 32, dtype=torch.int32, device=codes.device)
            canon, mate = pairs if pairs is not None else (None, None)
            _lib.check(L.sgs_edge_score_fwd_mask(_ptr(codes, torch.float32), _ptr(U, torch.float32), N, H, _ptr(edge_index, torch.int64), E,
                                                 edge_id_offset, _ptr(canon, torch.int32), 0 if canon is None else canon.numel(),
                                                 _ptr(mate, torch.int32), _ptr(W1, torch.float32), _ptr(b1), _ptr(w2), _ptr(b2), float(p), seed, site,
                                                 _ptr(out), _ptr(maskbits), ws.data_ptr(), ws.numel(), _stream()), "sgs_edge_score_fwd_mask")
        elif pairs is not None and E >= 65536 and L.sgs_edge_score_paired_supported(H):
            # undirected graph stored both ways: the canonical half of the edges runs the contraction, every mate rides along
            canon, mate = pairs
            _lib.check(L.sgs_edge_score_fwd_paired(_ptr(codes, torch.float32), _ptr(U, torch.float32), N, H, _ptr(edge_index, torch.int64), E,
                                                   edge_id_offset, _ptr(canon, torch.int32), canon.numel(), _ptr(mate, torch.int32),
                                                   _ptr(W1, torch.float32), _ptr(b1), _ptr(w2), _ptr(b2), float(p), seed, site, _ptr(out),
                                                   ws.data_ptr(), ws.numel(), _stream()), "sgs_edge_score_fwd_paired")
        else:
            _lib.check(L.sgs_edge_score_fwd(_ptr(codes, torch.float32), _ptr(U, torch.float32), N, H, _ptr(edge_index, torch.int64), E,
                                            edge_id_offset, _ptr(W1, torch.float32), _ptr(b1), _ptr(w2), _ptr(b2), float(p), seed, site, _ptr(out),
                                            ws.data_ptr(), ws.numel(), _stream()), "sgs_edge_score_fwd")
        ctx.save_for_backward(codes, U, W1, b1, w2, b2, edge_index, *((maskbits, out) if maskbits is not None else ()))
        ctx.active, ctx.p, ctx.seed, ctx.site, ctx.offset = active, float(p), seed, site, edge_id_offset
        # the fused backward needs the active rows grouped by source: true for a drawn subset (ascending ids) of a row-sorted list
        ctx.src_sorted = maskbits is not None and _fused_backward and src_sorted(edge_index)
        return out

    @staticmethod
    def backward(ctx, gp):
        L = _lib.lib()
        codes, U, W1, b1, w2, b2, edge_index = ctx.saved_tensors[:7]
        kept = ctx.saved_tensors[7:]                  # (maskbits, p) when the forward kept the mask
        N, H = codes.shape
        E = edge_index.shape[1]
        dev = codes.device
        act = ctx.active
        if act is not None and act.eid is not None:
            eid, graph = act.eid, act.graph
            n = eid.numel()
            tok = _zero_token(dev)
            if act.gq is not None and gp.numel() == E and gp.stride(0) == 0 and gp.data_ptr() == tok.data_ptr():
                gp_act = act.gq                     # handed over by _SelectSampled: `gp` is the stride-0 zero, nothing to gather
            else:
                gp_act = gp.index_select(0, eid)
                if act.gq is not None:              # another consumer of the scores contributed a dense gradient as well
                    gp_act = gp_act + act.gq
            act.gq = None
        else:
            eid, graph, n = None, get_graph(edge_index, N), E
            gp_act = gp.contiguous()
        f32 = dict(dtype=torch.float32, device=dev)
        if (_mask_backward and n >= 65536 and L.sgs_edge_score_bwd_bits_supported(H) and L.sgs_gemm_tn_mask_supported(n, H, H)
                and ctx.needs_input_grad[0]):
            return _EdgeScore._backward_mask(ctx, L, codes, U, W1, b1, w2, b2, edge_index, eid, graph, n, gp_act, kept)
        # (a kept mask goes unused when the active set turns out too small for the mask-form kernels: the dense path recomputes)
        dv, feat = torch.empty(n, H, **f32), torch.empty(n, H, **f32)
        tile = L.sgs_edge_score_bwd_tile()
        hdz = torch.empty((n + tile - 1) // tile, H, **f32)          # per-tile column sums of dz * hidden (rows sum to d w2)
        dz = torch.empty(n, **f32)
        if n > 0:
            ws = workspace(L.sgs_edge_score_workspace_bytes(N, H, 0), dev)
            _lib.check(L.sgs_edge_score_bwd_core(_ptr(codes), _ptr(U), N, H, _ptr(edge_index), E, ctx.offset, _ptr(eid), n, _ptr(gp_act),
                                                 _ptr(W1), _ptr(b1), _ptr(w2), _ptr(b2), ctx.p, ctx.seed, ctx.site, _ptr(dv),
                                                 _ptr(hdz), _ptr(dz), _ptr(feat), ws.data_ptr(), ws.numel(), _stream()),
                       "sgs_edge_score_bwd_core")
        if n >= 65536 and L.sgs_edge_score_bwd_dfeat_supported(H):
            # dfeat = dv W1a on the forward's bf16x6 loop as a row GEMM (fp32-faithful): ~75 us at 100 k rows against 122 us below
            dfeat = torch.empty(n, H, **f32)
            wsd = workspace(L.sgs_edge_score_workspace_bytes(0, H, 0), dev)
            _lib.check(L.sgs_edge_score_bwd_dfeat(_ptr(dv), n, H, _ptr(W1), _ptr(dfeat), wsd.data_ptr(), wsd.numel(), _stream()),
                       "sgs_edge_score_bwd_dfeat")
        else:
            # as F.linear with a contiguous W1a^T: the vendor GEMM runs that form at 110 TFLOP/s (85 with the strided view)
            W1a_t = W1[:, :H].t().contiguous()
            dfeat = torch.nn.functional.linear(dv, W1a_t)          # [n,H]
        # d fc1.weight [H, 2H], both halves written in place.  Left: dW1a = dv^T feat (K = n rows): sgs_gemm_tn's tall-K kernel (the
        # vendor GEMM picks a 42 TFLOP/s kernel for this shape), with d b1 = colsum(dv) as a by-product of the same pass over dv
        dW1 = torch.empty_like(W1)
        wsg = workspace(L.sgs_gemm_tn_workspace_bytes(n, H, H), dev)
        if L.sgs_gemm_tn_can_colsum(n, H, H):
            db1 = torch.empty(H, dtype=torch.float32, device=dev)
            _lib.check(L.sgs_gemm_tn_ld(_ptr(dv), _ptr(feat), n, H, H, _ptr(dW1), 2 * H, _ptr(db1), wsg.data_ptr(), wsg.numel(), _stream()),
                       "sgs_gemm_tn_ld")
        else:
            _lib.check(L.sgs_gemm_tn_ld(_ptr(dv), _ptr(feat), n, H, H, _ptr(dW1), 2 * H, None, wsg.data_ptr(), wsg.numel(), _stream()),
                       "sgs_gemm_tn_ld")
            db1 = _colsum(dv)
        dw2 = _colsum(hdz)
        db2 = _colsum(dz.view(n, 1)).reshape(1)
        if H % 4 == 0:                                         # both endpoint reductions in one pass over the incident-edge lists
            dcodes = torch.empty(N, H, dtype=torch.float32, device=dev)
            dU = torch.empty(N, H, dtype=torch.float32, device=dev)
            _lib.check(L.sgs_endpoint_reduce_pair(_ptr(dfeat), _ptr(dv), _ptr(codes), N, H, graph.n_edges, _ptr(graph.in_ptr), _ptr(graph.in_src),
                                                  _ptr(graph.in_eid), _ptr(graph.out_ptr), _ptr(graph.out_dst), _ptr(graph.out_eid),
                                                  _ptr(dcodes), _ptr(dU), _stream()), "sgs_endpoint_reduce_pair")
        else:
            dcodes = _endpoint_reduce(dfeat, dfeat, codes, graph, 1.0, 1.0, H)
            dU = _endpoint_reduce(dv, dv, None, graph, 1.0, -1.0, H)
        # the node-level half: U = codes W1b^T  ->  d codes += dU W1b (library GEMM, accumulating),  d W1b = dU^T codes (right half)
        if ctx.needs_input_grad[0]:
            dcodes.addmm_(dU, W1[:, H:])                           # in place (beta = 1): no copy of dcodes
        wsb = workspace(L.sgs_gemm_tn_workspace_bytes(N, H, H), dev)
        _lib.check(L.sgs_gemm_tn_ld(_ptr(dU), _ptr(codes), N, H, H, dW1.data_ptr() + 4 * H, 2 * H, None, wsb.data_ptr(), wsb.numel(), _stream()),
                   "sgs_gemm_tn_ld")
        return dcodes, dW1, db1, dw2, db2, None, None, None, None, None, None, None


def _variant_overrides_are_default() -> bool:
    """True while the library's forward / backward variant overrides are at their defaults (sgs_edge_score_set_variant(-1),
    sgs_edge_score_set_bwd_variant(-1)): only then may the mask-keeping forward (always the bf16x6 loop) stand in for the kernels a test or
    the bench asked for by name."""
    L = _lib.lib()
    return L.sgs_edge_score_get_variant() < 0 and L.sgs_edge_score_get_bwd_variant() < 0


def _edge_score_backward_mask(ctx, L, codes, U, W1, b1, w2, b2, edge_index, eid, graph, n, gp_act, kept=()):
    """The backward at production size in its MASK form (include/sgs_hip.h, sgs_edge_score_bwd_core_bits): dv = dz x [hidden > 0] x w2 / (1 - p)
    never exists as an fp32 [n, H] matrix -- the core writes one bit per entry and the three consumers rebuild what they need, the two
    contractions with a 0 / 1 operand at half the MFMA work."""
    N, H = codes.shape
    E = edge_index.shape[1]
    dev = codes.device
    f32 = dict(dtype=torch.float32, device=dev)
    p = ctx.p
    bits = torch.empty(n, H // 32, dtype=torch.int32, device=dev)
    dz = torch.empty(n, **f32)
    if kept and getattr(ctx, "src_sorted", False) and _fused_backward:      # (eid None: every edge active, in edge order -- sorted by source too)
        return _edge_score_backward_fused(ctx, L, codes, U, W1, b1, w2, edge_index, eid, graph, n, gp_act, kept, bits, dz)
    feat = torch.empty(n, H, **f32)
    hdz = Traw = craw = Rraw = None
    if kept:
        # the forward kept the mask and p: no recompute -- dz, the active rows' mask and feat in one pass; d fc2.weight from the consumers' parts
        maskbits, p_out = kept
        _lib.check(L.sgs_edge_score_bwd_prep(_ptr(codes), N, H, _ptr(edge_index), E, _ptr(eid), n, _ptr(gp_act), _ptr(p_out), _ptr(maskbits),
                                             _ptr(dz), _ptr(bits), _ptr(feat), _stream()), "sgs_edge_score_bwd_prep")
        Traw, craw, Rraw = torch.empty(H, H, **f32), torch.empty(H, **f32), torch.empty(N, H, **f32)
    else:
        tile = L.sgs_edge_score_bwd_tile()
        hdz = torch.empty((n + tile - 1) // tile, H, **f32)
        ws = workspace(L.sgs_edge_score_workspace_bytes(N, H, 0), dev)
        _lib.check(L.sgs_edge_score_bwd_core_bits(_ptr(codes), _ptr(U), N, H, _ptr(edge_index), E, ctx.offset, _ptr(eid), n, _ptr(gp_act), _ptr(W1),
                                                  _ptr(b1), _ptr(w2), _ptr(b2), p, ctx.seed, ctx.site, _ptr(bits), _ptr(hdz), _ptr(dz), _ptr(feat),
                                                  ws.data_ptr(), ws.numel(), _stream()), "sgs_edge_score_bwd_core_bits")
    dfeat = torch.empty(n, H, **f32)
    wsd = workspace(L.sgs_edge_score_workspace_bytes(0, H, 0), dev)
    _lib.check(L.sgs_edge_score_bwd_dfeat_bits(_ptr(bits), _ptr(dz), n, H, _ptr(W1), _ptr(w2), p, _ptr(dfeat), wsd.data_ptr(), wsd.numel(),
                                               _stream()), "sgs_edge_score_bwd_dfeat_bits")
    dW1 = torch.empty_like(W1)
    db1 = torch.empty(H, **f32)
    scale = float(np.float32(1.0) / (np.float32(1.0) - np.float32(p)))        # as the kernels form it: 1.0f / (1.0f - p)
    wsg = workspace(L.sgs_gemm_tn_workspace_bytes(n, H, H), dev)
    db2 = torch.empty(1, **f32)
    _lib.check(L.sgs_gemm_tn_mask(_ptr(bits), _ptr(dz), _ptr(w2), scale, _ptr(feat), n, H, H, _ptr(dW1), 2 * H, _ptr(db1), _ptr(db2), _ptr(Traw),
                                  _ptr(craw), wsg.data_ptr(), wsg.numel(), _stream()), "sgs_gemm_tn_mask")
    dw2 = _colsum(hdz) if hdz is not None else None
    dcodes = torch.empty(N, H, **f32)
    dU = torch.empty(N, H, **f32)
    _lib.check(L.sgs_endpoint_reduce_pair_bits(_ptr(dfeat), _ptr(bits), _ptr(dz), _ptr(w2), p, _ptr(codes), N, H, graph.n_edges, _ptr(graph.in_ptr),
                                               _ptr(graph.in_src), _ptr(graph.in_eid), _ptr(graph.out_ptr), _ptr(graph.out_dst),
                                               _ptr(graph.out_eid), _ptr(dcodes), _ptr(dU), _ptr(Rraw), _stream()), "sgs_endpoint_reduce_pair_bits")
    if dw2 is None:
        dw2 = torch.empty(H, **f32)
        _lib.check(L.sgs_edge_score_dw2_from_parts(_ptr(W1), _ptr(Traw), _ptr(U), _ptr(Rraw), _ptr(b1), _ptr(craw), N, H, p, _ptr(dw2), _stream()),
                   "sgs_edge_score_dw2_from_parts")
    return _edge_score_backward_mask_tail(L, codes, W1, dcodes, dU, dW1, db1, dw2, db2)


def _edge_score_backward_fused(ctx, L, codes, U, W1, b1, w2, edge_index, eid, graph, n, gp_act, kept, bits, dz):
    """The no-recompute backward with neither feat nor dfeat as [n, H] arrays (include/sgs_hip.h, "FUSED form"): the active rows are sorted
    by source, so the by-source half of d codes is reduced inside the dfeat contraction's epilogue, and the weight-gradient GEMM gathers
    x_s * x_d itself.  Four launches (+ the W1a pack): prep (dz, mask rows, endpoints), dfeat + by-source sums, d W1a, the reductions."""
    N, H = codes.shape
    E = edge_index.shape[1]
    dev = codes.device
    f32 = dict(dtype=torch.float32, device=dev)
    p = ctx.p
    maskbits, p_out = kept
    sd = torch.empty(n, 2, dtype=torch.int32, device=dev)
    _lib.check(L.sgs_edge_score_bwd_prep_sd(_ptr(codes), N, H, _ptr(edge_index), E, _ptr(eid), n, _ptr(gp_act), _ptr(p_out), _ptr(maskbits),
                                            _ptr(dz), _ptr(bits), _ptr(sd), _stream()), "sgs_edge_score_bwd_prep_sd")
    G = torch.empty(n, H, **f32)
    opart = torch.empty(L.sgs_edge_score_bwd_fused_opart_rows(n, N), H, **f32)
    wsd = workspace(L.sgs_edge_score_workspace_bytes(0, H, 0), dev)
    _lib.check(L.sgs_edge_score_bwd_dfeat_fused(_ptr(bits), _ptr(dz), _ptr(sd), _ptr(codes), n, N, H, _ptr(W1), _ptr(w2), p, _ptr(G), _ptr(opart),
                                                wsd.data_ptr(), wsd.numel(), _stream()), "sgs_edge_score_bwd_dfeat_fused")
    dW1 = torch.empty_like(W1)
    db1, db2 = torch.empty(H, **f32), torch.empty(1, **f32)
    Traw, craw, Rraw = torch.empty(H, H, **f32), torch.empty(H, **f32), torch.empty(N, H, **f32)
    scale = float(np.float32(1.0) / (np.float32(1.0) - np.float32(p)))
    wsg = workspace(L.sgs_gemm_tn_workspace_bytes(n, H, H), dev)
    _lib.check(L.sgs_gemm_tn_mask_gather(_ptr(bits), _ptr(dz), _ptr(w2), scale, _ptr(codes), N, _ptr(sd), n, H, H, _ptr(dW1), 2 * H, _ptr(db1), _ptr(db2),
                                         _ptr(Traw), _ptr(craw), wsg.data_ptr(), wsg.numel(), _stream()), "sgs_gemm_tn_mask_gather")
    dcodes, dU = torch.empty(N, H, **f32), torch.empty(N, H, **f32)
    _lib.check(L.sgs_edge_score_bwd_reduce_fused(_ptr(G), _ptr(opart), _ptr(bits), _ptr(dz), _ptr(w2), p, N, H, graph.n_edges, _ptr(graph.in_ptr),
                                                 _ptr(graph.in_eid), _ptr(graph.out_ptr), _ptr(dcodes), _ptr(dU), _ptr(Rraw), _stream()),
               "sgs_edge_score_bwd_reduce_fused")
    dw2 = torch.empty(H, **f32)
    _lib.check(L.sgs_edge_score_dw2_from_parts(_ptr(W1), _ptr(Traw), _ptr(U), _ptr(Rraw), _ptr(b1), _ptr(craw), N, H, p, _ptr(dw2), _stream()),
               "sgs_edge_score_dw2_from_parts")
    return _edge_score_backward_mask_tail(L, codes, W1, dcodes, dU, dW1, db1, dw2, db2)


def _edge_score_backward_mask_tail(L, codes, W1, dcodes, dU, dW1, db1, dw2, db2):
    """The node-level half: U = codes W1b^T  ->  d codes += dU W1b,  d W1b = dU^T codes (right half of d fc1.weight, in place)."""
    N, H = codes.shape
    dcodes.addmm_(dU, W1[:, H:])
    wsb = workspace(L.sgs_gemm_tn_workspace_bytes(N, H, H), codes.device)
    _lib.check(L.sgs_gemm_tn_ld(_ptr(dU), _ptr(codes), N, H, H, dW1.data_ptr() + 4 * H, 2 * H, None, wsb.data_ptr(), wsb.numel(), _stream()),
               "sgs_gemm_tn_ld")
    return dcodes, dW1, db1, dw2, db2, None, None, None, None, None, None, None


_EdgeScore._backward_mask = staticmethod(_edge_score_backward_mask)


def edge_score(codes, fc1_w, fc1_b, fc2_w, fc2_b, edge_index, active=None, p=0.0, seed=0, site=0, edge_id_offset=0, pairs="cached"):
    """K1b.  codes [N,H]; fc1_w [H,2H]; fc1_b [H]; fc2_w [1,H]; fc2_b [1]; edge_index [2,E] -> p [E].
    `pairs`: (canon, mate) of get_pairs for the paired forward, None for the plain one, "cached" (default) = whatever
    get_pairs(edge_index) holds (nothing is built here)."""
    _need_gpu(codes, fc1_w, edge_index)
    if isinstance(pairs, str):
        pairs = get_pairs(edge_index, codes.shape[0])
    return _EdgeScore.apply(codes.contiguous(), fc1_w.contiguous(), fc1_b.contiguous(), fc2_w.reshape(-1).contiguous(), fc2_b.contiguous(),
                            edge_index.contiguous(), active, float(p), int(seed), int(site), int(edge_id_offset), pairs)


class _EdgeScoreEPD(torch.autograd.Function):
    """EdgeProbMLP with dropout > 0 (model.py:16-45): `_edge_score(drop(A[src]), drop(A[dst]))` with the two endpoint masks drawn per (edge,
    endpoint), straight from the node table A = relu(fcdim(X)) [N, H] and the scored edge list -- no [E', H] gathers, masks or
    concatenations on the way (include/sgs_hip.h, "Endpoint-dropout scorer").  The backward runs over the active rows only (the trainer's
    ActiveSet: the q sampled edges in the hybrid pipeline) and materialises the features of THOSE rows alone."""

    @staticmethod
    def forward(ctx, A, W1, b1, w2, b2, edge_index, active, p, seed, site, p_ep, seed_x, site_x, seed_y, site_y, edge_id_offset):
        L = _lib.lib()
        N, H = A.shape
        E = edge_index.shape[1]
        out = torch.empty(E, dtype=torch.float32, device=A.device)
        ws = workspace(L.sgs_edge_score_epd_workspace_bytes(H), A.device)
        _lib.check(L.sgs_edge_score_epd_fwd(_ptr(A, torch.float32), N, H, _ptr(edge_index, torch.int64), E, edge_id_offset, _ptr(W1, torch.float32),
                                            _ptr(b1), _ptr(w2), _ptr(b2), float(p), seed, site, float(p_ep), seed_x, site_x, seed_y, site_y, _ptr(out),
                                            ws.data_ptr(), ws.numel(), _stream()), "sgs_edge_score_epd_fwd")
        ctx.save_for_backward(A, W1, b1, w2, b2, edge_index)
        ctx.active, ctx.cfg = active, (float(p), seed, site, float(p_ep), seed_x, site_x, seed_y, site_y, edge_id_offset)
        return out

    @staticmethod
    def backward(ctx, gp):
        L = _lib.lib()
        A, W1, b1, w2, b2, edge_index = ctx.saved_tensors
        p, seed, site, p_ep, seed_x, site_x, seed_y, site_y, offset = ctx.cfg
        N, H = A.shape
        E = edge_index.shape[1]
        dev = A.device
        act = ctx.active
        if act is not None and act.eid is not None:
            eid, graph = act.eid, act.graph
            n = eid.numel()
            tok = _zero_token(dev)
            if act.gq is not None and gp.numel() == E and gp.stride(0) == 0 and gp.data_ptr() == tok.data_ptr():
                gp_act = act.gq
            else:
                gp_act = gp.index_select(0, eid)
                if act.gq is not None:
                    gp_act = gp_act + act.gq
            act.gq = None
        else:
            eid, graph, n = None, get_graph(edge_index, N), E
            gp_act = gp.contiguous()
        f32 = dict(dtype=torch.float32, device=dev)
        dv, feat2, dz = torch.empty(n, H, **f32), torch.empty(n, 2 * H, **f32), torch.empty(n, **f32)
        tile = L.sgs_edge_score_bwd_tile()
        hdz = torch.empty((n + tile - 1) // tile, H, **f32)
        dA = torch.zeros(N, H, **f32) if n == 0 else torch.empty(N, H, **f32)
        dW1 = torch.zeros_like(W1) if n == 0 else torch.empty_like(W1)
        if n == 0:
            return dA, dW1, torch.zeros(H, **f32), torch.zeros(H, **f32), torch.zeros(1, **f32), *([None] * 11)
        ws = workspace(L.sgs_edge_score_epd_workspace_bytes(H), dev)
        _lib.check(L.sgs_edge_score_epd_bwd_core(_ptr(A), N, H, _ptr(edge_index), E, offset, _ptr(eid), n, _ptr(gp_act), _ptr(W1), _ptr(b1), _ptr(w2),
                                                 _ptr(b2), p, seed, site, p_ep, seed_x, site_x, seed_y, site_y, _ptr(dv), _ptr(hdz), _ptr(dz), _ptr(feat2),
                                                 ws.data_ptr(), ws.numel(), _stream()), "sgs_edge_score_epd_bwd_core")
        wsg = workspace(L.sgs_gemm_tn_workspace_bytes(n, H, 2 * H), dev)
        _lib.check(L.sgs_gemm_tn(_ptr(dv), _ptr(feat2), n, H, 2 * H, _ptr(dW1), wsg.data_ptr(), wsg.numel(), _stream()), "sgs_gemm_tn")
        db1, dw2 = _colsum(dv), _colsum(hdz)
        db2 = _colsum(dz.view(n, 1)).reshape(1)
        dfeat2 = torch.mm(dv, W1)                               # [n, 2H] = [d (x_m * y_m) | d (x_m - y_m)]  (library GEMM)
        _lib.check(L.sgs_edge_score_epd_reduce(_ptr(dfeat2), _ptr(A), N, H, _ptr(graph.in_ptr), _ptr(graph.in_src), _ptr(graph.in_eid), _ptr(graph.out_ptr),
                                               _ptr(graph.out_dst), _ptr(graph.out_eid), _ptr(eid), offset, p_ep, seed_x, site_x, seed_y, site_y, _ptr(dA),
                                               _stream()), "sgs_edge_score_epd_reduce")
        return dA, dW1, db1, dw2, db2, *([None] * 11)


def edge_score_epd(A, fc1_w, fc1_b, fc2_w, fc2_b, edge_index, active=None, p=0.0, seed=0, site=0, p_ep=0.0, seed_x=0, site_x=0, seed_y=0,
                   site_y=0, edge_id_offset=0):
    """The scorer of EdgeProbMLP when its endpoint dropout is on (see _EdgeScoreEPD): A [N,H] = relu(fcdim(X)); -> p [E]."""
    _need_gpu(A, fc1_w, edge_index)
    return _EdgeScoreEPD.apply(A.contiguous(), fc1_w.contiguous(), fc1_b.contiguous(), fc2_w.reshape(-1).contiguous(), fc2_b.contiguous(),
                               edge_index.contiguous(), active, float(p), int(seed), int(site), float(p_ep), int(seed_x), int(site_x), int(seed_y),
                               int(site_y), int(edge_id_offset))


# ------------------------------------------------------------------ gate + losses (K6)
def _u8(mask: torch.Tensor) -> torch.Tensor:
    if mask.dtype == torch.bool:
        return mask.contiguous().view(torch.uint8)
    if mask.dtype == torch.uint8:
        return mask.contiguous()
    raise RuntimeError("mask must be a bool tensor")


def masked_correct(logits, y, train_mask, out=None) -> torch.Tensor:
    """int32 [2] on device: (#correct argmax on train rows, #train rows) -- no host sync."""
    L = _lib.lib()
    _need_gpu(logits, y, train_mask)
    if out is None:
        out = torch.empty(2, dtype=torch.int32, device=logits.device)
    N, C = logits.shape
    _lib.check(L.sgs_masked_correct(_ptr(logits.contiguous(), torch.float32), N, C, _ptr(y, torch.int64), _ptr(_u8(train_mask)),
                                    _ptr(out), _stream()), "sgs_masked_correct")
    return out


def masked_correct_pair(logits_a, logits_b, y, train_mask, out) -> torch.Tensor:
    """The gate's two counts in one launch: out[0:4] = (#correct_a, #train, #correct_b, #train); `out` (int32, >= 4 entries)
    must be zero on entry."""
    L = _lib.lib()
    _need_gpu(logits_a, logits_b, y, train_mask, out)
    N, C = logits_a.shape
    if logits_b.shape != logits_a.shape:
        raise RuntimeError("masked_correct_pair: logits shapes differ")
    _lib.check(L.sgs_masked_correct_pair(_ptr(logits_a.contiguous(), torch.float32), _ptr(logits_b.contiguous(), torch.float32), N, C,
                                         _ptr(y, torch.int64), _ptr(_u8(train_mask)), _ptr(out, torch.int32), _stream()),
               "sgs_masked_correct_pair")
    return out


def gate_counts(logits_a, logits_b, y, train_mask, publish=None) -> torch.Tensor:
    """The gate's counts: int32[5] on the device = (#correct_a, #train, #correct_b, #train, 0), two launches, no zero fill.
    `publish` = (seq, dst_pinned): the finishing launch also hands the four counts to the host (as publish_to_host)."""
    L = _lib.lib()
    _need_gpu(logits_a, logits_b, y, train_mask)
    N, C = logits_a.shape
    if logits_b.shape != logits_a.shape:
        raise RuntimeError("gate_counts: logits shapes differ")
    out = torch.empty(5, dtype=torch.int32, device=logits_a.device)
    seq_p = dst_p = None
    if publish is not None:
        seq, dst = publish
        if dst.is_cuda or not dst.is_pinned() or dst.dtype != torch.int32 or dst.numel() < 5:
            raise RuntimeError("gate_counts: the publish destination must be a pinned host int32 tensor with 5 entries")
        seq_p, dst_p = (None if seq is None else seq.data_ptr()), dst.data_ptr()
    ws = workspace(L.sgs_gate_counts_workspace_bytes(N), logits_a.device)
    _lib.check(L.sgs_gate_counts(_ptr(logits_a.contiguous(), torch.float32), _ptr(logits_b.contiguous(), torch.float32), N, C, _ptr(y, torch.int64),
                                 _ptr(_u8(train_mask)), _ptr(out, torch.int32), seq_p, dst_p, ws.data_ptr(), ws.numel(), _stream()),
               "sgs_gate_counts")
    return out


def loss_tick(loss_sum, loss, epoch) -> None:
    """sgs_loss_tick: loss_sum += loss and epoch += 1 in one launch (the last kernel of a replayed step)."""
    L = _lib.lib()
    _need_gpu(loss_sum, loss, epoch)
    _lib.check(L.sgs_loss_tick(_ptr(loss_sum, torch.float32), _ptr(loss.detach().reshape(1), torch.float32), epoch.data_ptr(), _stream()),
               "sgs_loss_tick")


def publish_to_host(src, n, seq, dst_pinned) -> None:
    """sgs_publish_to_host: src (device int32) -> dst_pinned (pinned host int32, >= n + 1 entries); seq: device int64 word."""
    L = _lib.lib()
    _need_gpu(src, seq)
    if dst_pinned.is_cuda or not dst_pinned.is_pinned() or dst_pinned.dtype != torch.int32 or dst_pinned.numel() < n + 1:
        raise RuntimeError("publish_to_host: destination must be a pinned host int32 tensor with n + 1 entries")
    _lib.check(L.sgs_publish_to_host(_ptr(src, torch.int32), n, None if seq is None else seq.data_ptr(), dst_pinned.data_ptr(), _stream()),
               "sgs_publish_to_host")


class _MaskedCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, y, mask_u8):
        L = _lib.lib()
        N, C = logits.shape
        dev = logits.device
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        row_lse = torch.empty(N, dtype=torch.float32, device=dev)
        rowloss = torch.empty(N, dtype=torch.float32, device=dev)
        n_rows = torch.empty(1, dtype=torch.int32, device=dev)
        _lib.check(L.sgs_masked_ce_fwd(_ptr(logits), N, C, _ptr(y), _ptr(mask_u8), _ptr(loss), _ptr(row_lse), _ptr(rowloss),
                                       _ptr(n_rows), _stream()), "sgs_masked_ce_fwd")
        ctx.save_for_backward(logits, y, mask_u8, row_lse, n_rows)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        logits, y, mask_u8, row_lse, n_rows = ctx.saved_tensors
        N, C = logits.shape
        g = g.reshape(1).contiguous().float()
        d = torch.empty_like(logits)
        _lib.check(L.sgs_masked_ce_bwd(_ptr(logits), N, C, _ptr(y), _ptr(mask_u8), _ptr(row_lse), _ptr(n_rows), _ptr(g), _ptr(d),
                                       _stream()), "sgs_masked_ce_bwd")
        return d, None, None


def masked_cross_entropy(logits, y, train_mask):
    """nn.CrossEntropyLoss()(logits[train_mask], y[train_mask]) without the boolean-index sync."""
    _need_gpu(logits, y, train_mask)
    return _MaskedCE.apply(logits.contiguous(), y.contiguous(), _u8(train_mask))


class _EdgeReg(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w, logits, sei, y, mask_u8, graph, coef1, coef2, box):
        L = _lib.lib()
        q = w.numel()
        N, C = logits.shape
        dev = w.device
        out = torch.empty(5, dtype=torch.float32, device=dev)
        ws = workspace(L.sgs_edge_reg_workspace_bytes(q), dev)
        _lib.check(L.sgs_edge_reg_fwd(_ptr(w), _ptr(sei), q, _ptr(logits), N, C, _ptr(y), _ptr(mask_u8), float(coef1), float(coef2),
                                      _ptr(out), None, ws.data_ptr(), ws.numel(), _stream()), "sgs_edge_reg_fwd")
        ctx.save_for_backward(w, logits, sei, y, mask_u8, out)
        ctx.graph, ctx.coef1, ctx.coef2 = graph, float(coef1), float(coef2)
        box.append(out)
        return out[4]              # 0-dim view of the saved [5] vector (no copy launch)

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        w, logits, sei, y, mask_u8, out = ctx.saved_tensors
        q = w.numel()
        N, C = logits.shape
        dev = w.device
        g = g.reshape(1).contiguous().float()
        dw = torch.empty(q, dtype=torch.float32, device=dev)
        Gs = torch.empty(q, C, dtype=torch.float32, device=dev)
        Gd = torch.empty(q, C, dtype=torch.float32, device=dev)
        _lib.check(L.sgs_edge_reg_bwd(_ptr(w), _ptr(sei), q, q, _ptr(logits), N, C, _ptr(y), _ptr(mask_u8), _ptr(out), ctx.coef1,
                                      ctx.coef2, _ptr(g), _ptr(dw), _ptr(Gs), _ptr(Gd), _stream()), "sgs_edge_reg_bwd")
        dlogits = _endpoint_reduce(Gs, Gd, None, ctx.graph, 1.0, 1.0, C) if ctx.coef2 != 0.0 else None
        return dw, dlogits, None, None, None, None, None, None, None


def edge_regularizers(w, logits, sampled_edge_index, y, train_mask, coef1, coef2):
    """coef1 * reg1 + coef2 * reg2 (training_hybrid.py:107-133) as one scalar, plus the detached
    [reg1, reg2, #valid, sum labels, total] vector."""
    _need_gpu(w, logits, sampled_edge_index, y, train_mask)
    graph = get_graph(sampled_edge_index, logits.shape[0])
    box = []
    total = _EdgeReg.apply(w.contiguous(), logits.contiguous(), sampled_edge_index.contiguous(), y.contiguous(),
                           _u8(train_mask), graph, float(coef1), float(coef2), box)
    return total, box[0]


class _HybridLoss(torch.autograd.Function):
    """criterion + coef1 reg1 + coef2 reg2 as ONE node: three launches forward (row losses, per-edge terms, one finishing block),
    three backward (per-edge gradients, their endpoint reduction, the cross entropy's gradient added in place) -- as separate nodes
    the same sum took ten, two of them the adds autograd inserts."""

    @staticmethod
    def forward(ctx, logits, y, mask_u8, w, sei, graph, coef1, coef2, box):
        L = _lib.lib()
        q = w.numel()
        N, C = logits.shape
        dev = w.device
        out = torch.empty(7, dtype=torch.float32, device=dev)
        row_lse = torch.empty(N, dtype=torch.float32, device=dev)
        rowloss = torch.empty(N, dtype=torch.float32, device=dev)
        n_rows = torch.empty(1, dtype=torch.int32, device=dev)
        ws = workspace(L.sgs_edge_reg_workspace_bytes(q), dev)
        _lib.check(L.sgs_hybrid_loss_fwd(_ptr(logits), N, C, _ptr(y), _ptr(mask_u8), _ptr(w), _ptr(sei), q, float(coef1), float(coef2), _ptr(out),
                                         _ptr(row_lse), _ptr(rowloss), _ptr(n_rows), ws.data_ptr(), ws.numel(), _stream()), "sgs_hybrid_loss_fwd")
        ctx.save_for_backward(logits, y, mask_u8, w, sei, out, row_lse, n_rows)
        ctx.graph, ctx.coef1, ctx.coef2 = graph, float(coef1), float(coef2)
        ctx.nm_ref = getattr(w, "_sgs_norm", None)         # the normalisation that differentiates these weights, if any (note_first_dw)
        box.append(out)
        return out[6]

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        logits, y, mask_u8, w, sei, out, row_lse, n_rows = ctx.saved_tensors
        q = w.numel()
        N, C = logits.shape
        dev = w.device
        g = g.reshape(1).contiguous().float()
        dw = torch.empty(q, dtype=torch.float32, device=dev)
        Gs = torch.empty(q, C, dtype=torch.float32, device=dev)
        Gd = torch.empty(q, C, dtype=torch.float32, device=dev)
        _lib.check(L.sgs_edge_reg_bwd(_ptr(w), _ptr(sei), q, q, _ptr(logits), N, C, _ptr(y), _ptr(mask_u8), _ptr(out), ctx.coef1,
                                      ctx.coef2, _ptr(g), _ptr(dw), _ptr(Gs), _ptr(Gd), _stream()), "sgs_edge_reg_bwd")
        if ctx.coef2 != 0.0:
            dlogits = _endpoint_reduce(Gs, Gd, None, ctx.graph, 1.0, 1.0, C)
            _lib.check(L.sgs_masked_ce_bwd_acc(_ptr(logits), N, C, _ptr(y), _ptr(mask_u8), _ptr(row_lse), _ptr(n_rows), _ptr(g), _ptr(dlogits),
                                               _stream()), "sgs_masked_ce_bwd_acc")
        else:
            dlogits = torch.empty_like(logits)
            _lib.check(L.sgs_masked_ce_bwd(_ptr(logits), N, C, _ptr(y), _ptr(mask_u8), _ptr(row_lse), _ptr(n_rows), _ptr(g), _ptr(dlogits),
                                           _stream()), "sgs_masked_ce_bwd")
        nm = ctx.nm_ref() if ctx.nm_ref is not None else None
        if nm is not None and getattr(nm, "_park_ok", False) and dw.numel() == nm.graph.n_edges:
            import weakref
            nm._dw_first = weakref.ref(dw)                 # the normalisation's backward accumulates into dw in place (no autograd add)
        return dlogits, None, None, dw, None, None, None, None, None


def hybrid_loss(logits, y, train_mask, w, sampled_edge_index, coef1, coef2):
    """nn.CrossEntropyLoss()(logits[train], y[train]) + coef1 * reg1 + coef2 * reg2 (training_hybrid.py:105-133) as one scalar, plus
    the detached [reg1, reg2, #valid, sum labels, coef1 reg1 + coef2 reg2, cross entropy, loss] vector."""
    _need_gpu(w, logits, sampled_edge_index, y, train_mask)
    graph = get_graph(sampled_edge_index, logits.shape[0])
    box = []
    total = _HybridLoss.apply(logits.contiguous(), y.contiguous(), _u8(train_mask), w.contiguous(), sampled_edge_index.contiguous(), graph,
                              float(coef1), float(coef2), box)
    return total, box[0]


# ------------------------------------------------------------------ GAT attention (K8)
class _GATAggregate(torch.autograd.Function):
    """out = act( sum_k alpha_k x'[src_k] + alpha_loop x'[i] + bias ), alpha = dropout(softmax(leaky_relu(a_s+a_d)))."""

    @staticmethod
    def forward(ctx, xl, a_s, a_d, bias, graph, slope, p_att, seed_att, site_att, act, p_act, seed_act, site_act):
        L = _lib.lib()
        N, D = xl.shape
        n = graph.n_edges
        dev = xl.device
        f32 = dict(dtype=torch.float32, device=dev)
        soft_in, alpha_in = torch.empty(max(n, 1), **f32), torch.empty(max(n, 1), **f32)
        soft_loop, alpha_loop = torch.empty(N, **f32), torch.empty(N, **f32)
        _lib.check(L.sgs_gat_alpha_fwd(_ptr(a_s), _ptr(a_d), N, n, _ptr(graph.in_ptr), _ptr(graph.in_src), _ptr(graph.in_eid),
                                       float(slope), float(p_att), seed_att, site_att, _ptr(soft_in), _ptr(soft_loop),
                                       _ptr(alpha_in), _ptr(alpha_loop), _stream()), "sgs_gat_alpha_fwd")
        Y = _spmm(xl, graph.in_ptr, graph.in_src, alpha_in, alpha_loop, bias, act, p_act, seed_act, site_act, N, D, n)
        ctx.save_for_backward(xl, a_s, a_d, soft_in, soft_loop, alpha_in, alpha_loop, Y if act != ACT_NONE else None)
        ctx.graph, ctx.slope, ctx.p_att, ctx.seed_att, ctx.site_att = graph, float(slope), float(p_att), seed_att, site_att
        ctx.act, ctx.p_act, ctx.has_bias = act, float(p_act), bias is not None
        return Y

    @staticmethod
    def backward(ctx, dY):
        L = _lib.lib()
        xl, a_s, a_d, soft_in, soft_loop, alpha_in, alpha_loop, Y = ctx.saved_tensors
        gr = ctx.graph
        N, D = xl.shape
        n = gr.n_edges
        dev = xl.device
        f32 = dict(dtype=torch.float32, device=dev)
        dY = dY.contiguous()
        dbias = None
        if ctx.act != ACT_NONE and ctx.has_bias:
            dZ, dbias = _act_bwd_colsum(dY, Y, ctx.act, ctx.p_act)
        elif ctx.act != ACT_NONE:
            dZ = torch.empty_like(dY)
            _lib.check(L.sgs_act_bwd(_ptr(dY), _ptr(Y), dY.numel(), ctx.act, ctx.p_act, _ptr(dZ), _stream()), "sgs_act_bwd")
        else:
            dZ = dY
            dbias = _colsum(dZ) if ctx.has_bias else None
        # alpha re-ordered into src-CSR entry order for the transposed aggregation
        by_eid = torch.empty(max(n, 1), **f32)
        alpha_out = torch.empty(max(n, 1), **f32)
        _lib.check(L.sgs_scatter_by_eid(_ptr(alpha_in), _ptr(gr.in_eid), n, _ptr(by_eid), _stream()), "sgs_scatter_by_eid")
        _lib.check(L.sgs_gather_by_eid(_ptr(by_eid), _ptr(gr.out_eid), n, _ptr(alpha_out), _stream()), "sgs_gather_by_eid")
        dxl = _spmm(dZ, gr.out_ptr, gr.out_dst, alpha_out, alpha_loop, None, ACT_NONE, 0.0, 0, 0, N, D, n)
        galpha, gloop = torch.empty(max(n, 1), **f32), torch.empty(N, **f32)
        _lib.check(L.sgs_sddmm_csr(_ptr(dZ), _ptr(xl), N, D, n, _ptr(gr.in_ptr), _ptr(gr.in_src), _ptr(gr.in_eid), _ptr(galpha),
                                   _ptr(gloop), _stream()), "sgs_sddmm_csr")
        g_edge, g_self, d_ad = torch.empty(max(n, 1), **f32), torch.empty(N, **f32), torch.empty(N, **f32)
        _lib.check(L.sgs_gat_alpha_bwd(_ptr(a_s), _ptr(a_d), N, n, _ptr(gr.in_ptr), _ptr(gr.in_src), _ptr(gr.in_eid), ctx.slope,
                                       ctx.p_att, ctx.seed_att, ctx.site_att, _ptr(soft_in), _ptr(soft_loop), _ptr(galpha),
                                       _ptr(gloop), _ptr(g_edge), _ptr(g_self), _ptr(d_ad), _stream()), "sgs_gat_alpha_bwd")
        g_out = torch.empty(max(n, 1), **f32)
        _lib.check(L.sgs_gather_by_eid(_ptr(g_edge), _ptr(gr.out_eid), n, _ptr(g_out), _stream()), "sgs_gather_by_eid")
        ones = torch.ones(N, 1, **f32)
        d_as = _spmm(ones, gr.out_ptr, gr.out_dst, g_out, g_self, None, ACT_NONE, 0.0, 0, 0, N, 1, n).reshape(N)
        return dxl, d_as, d_ad, dbias, None, None, None, None, None, None, None, None, None


class _GATScores(torch.autograd.Function):
    """a_s = x' att_src, a_d = x' att_dst (GATConv's node-level dots) in one pass over x' (sgs_gat_scores_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, xl, att_s, att_d):
        L = _lib.lib()
        N, D = xl.shape
        a_s, a_d = torch.empty(N, dtype=torch.float32, device=xl.device), torch.empty(N, dtype=torch.float32, device=xl.device)
        _lib.check(L.sgs_gat_scores_fwd(_ptr(xl, torch.float32), N, D, _ptr(att_s), _ptr(att_d), _ptr(a_s), _ptr(a_d), _stream()), "sgs_gat_scores_fwd")
        ctx.save_for_backward(xl, att_s, att_d)
        return a_s, a_d

    @staticmethod
    def backward(ctx, g_s, g_d):
        L = _lib.lib()
        xl, att_s, att_d = ctx.saved_tensors
        N, D = xl.shape
        dev = xl.device
        g_s = torch.zeros(N, dtype=torch.float32, device=dev) if g_s is None else g_s.contiguous()
        g_d = torch.zeros(N, dtype=torch.float32, device=dev) if g_d is None else g_d.contiguous()
        dxl = torch.empty_like(xl)
        das, dad = torch.empty(D, dtype=torch.float32, device=dev), torch.empty(D, dtype=torch.float32, device=dev)
        ws = workspace(L.sgs_gat_scores_bwd_workspace_bytes(N, D), dev)
        _lib.check(L.sgs_gat_scores_bwd(_ptr(xl), N, D, _ptr(att_s), _ptr(att_d), _ptr(g_s), _ptr(g_d), 0, _ptr(dxl), _ptr(das), _ptr(dad), ws.data_ptr(),
                                        ws.numel(), _stream()), "sgs_gat_scores_bwd")
        return dxl, das, dad


def gat_scores(xl, att_src, att_dst):
    _need_gpu(xl, att_src, att_dst)
    return _GATScores.apply(xl.contiguous(), att_src.reshape(-1).contiguous(), att_dst.reshape(-1).contiguous())


def gat_aggregate(xl, a_s, a_d, bias, graph: Graph, negative_slope=0.2, p_att=0.0, seed_att=0, site_att=0, act=ACT_NONE,
                  p_act=0.0, seed_act=0, site_act=0):
    _need_gpu(xl, a_s, a_d, bias)
    return _GATAggregate.apply(xl.contiguous(), a_s.contiguous(), a_d.contiguous(), bias, graph, float(negative_slope),
                               float(p_att), int(seed_att), int(site_att), act, float(p_act), int(seed_act), int(site_act))


# ------------------------------------------------------------------ node-level Linear with a hand-written weight gradient
class _LinearNoBias(torch.autograd.Function):
    """y = x W^T (library GEMM); dW = dY^T x on the f32 matrix cores (sgs_gemm_tn); dx = dY W (library)."""

    @staticmethod
    def forward(ctx, x, W):
        ctx.save_for_backward(x, W)
        return x @ W.t()                     # W may be a strided view (e.g. fc1.weight[:, H:])

    @staticmethod
    def backward(ctx, dY):
        L = _lib.lib()
        x, W = ctx.saved_tensors
        dx = dW = None
        dY = dY.contiguous()
        if ctx.needs_input_grad[0]:
            dx = dY @ W
        if ctx.needs_input_grad[1]:
            K, M, N = x.shape[0], W.shape[0], W.shape[1]
            dW = torch.empty(M, N, dtype=torch.float32, device=x.device)
            ws = workspace(L.sgs_gemm_tn_workspace_bytes(K, M, N), x.device)
            _lib.check(L.sgs_gemm_tn(_ptr(dY, torch.float32), _ptr(x.contiguous(), torch.float32), K, M, N, _ptr(dW), ws.data_ptr(),
                                     ws.numel(), _stream()), "sgs_gemm_tn")
        return dx, dW


def linear_nobias(x, W):
    _need_gpu(x, W)
    return _LinearNoBias.apply(x, W)


# ------------------------------------------------------------------ GraphSAGE mean aggregation, device-side degree prior
def mean_norm(graph: Graph) -> Norm:
    """Norm-like object for SAGEConv's mean aggregation: weights 1/indeg(dst), no self-loop term."""
    L = _lib.lib()
    dev = graph.edge_index.device
    nm = Norm()
    nm.graph, nm.w, nm.handle, nm.dis, nm.loopw, nm.what_loop = graph, None, None, None, None, None
    ne = max(graph.n_edges, 1)
    nm.what_in = torch.empty(ne, dtype=torch.float32, device=dev)
    nm.what_out = torch.empty(ne, dtype=torch.float32, device=dev)
    _lib.check(L.sgs_mean_weights(graph.n_edges, graph.N, _ptr(graph.in_ptr), _ptr(graph.out_ptr), _ptr(graph.out_dst),
                                  _ptr(nm.what_in), _ptr(nm.what_out), _stream()), "sgs_mean_weights")
    return nm


def sum_norm(graph: Graph, diag: float = 1.0) -> Norm:
    """Norm-like object for GINConv's sum aggregation: unit edge weights and `diag` = 1 + eps on the node itself."""
    nm = getattr(graph, "_norm_sum", None)
    if nm is not None and nm[0] == diag:
        return nm[1]
    dev = graph.edge_index.device
    nm = Norm()
    nm.graph, nm.w, nm.handle, nm.dis, nm.loopw = graph, None, None, None, None
    ne = max(graph.n_edges, 1)
    nm.what_in = torch.ones(ne, dtype=torch.float32, device=dev)
    nm.what_out = nm.what_in
    nm.what_loop = torch.full((max(graph.N, 1),), float(diag), dtype=torch.float32, device=dev)
    graph._norm_sum = (diag, nm)
    return nm


def degree_prior(edge_index: torch.Tensor, num_nodes: int) -> torch.Tensor:
    """`data.prob` of datasets.py:141-156 (add_degree) computed on the device."""
    L = _lib.lib()
    _need_gpu(edge_index)
    g = get_graph(edge_index, num_nodes)
    E = edge_index.shape[1]
    logits = torch.empty(E, dtype=torch.float32, device=edge_index.device)
    _lib.check(L.sgs_degree_prior_logits(_ptr(g.edge_index), E, num_nodes, _ptr(g.in_ptr), _ptr(g.out_ptr), _ptr(logits), _stream()),
               "sgs_degree_prior_logits")
    return torch.softmax(logits, dim=0)


def er_prior(edge_index: torch.Tensor, num_nodes: int, seed: int = 0, walk_lengths: int = 4, walks: int = 100, raw: bool = False) -> torch.Tensor:
    """`data.prob` of datasets.py:159-173 (add_ER) computed on the device: random-walk effective-resistance weights
    (sgs_er_weight) then softmax(weight * E^-1/2).  `edge_index` must be symmetric and coalesced (what the reference's
    to_networkx(to_undirected=True) walks on); raw=True returns the un-normalised weights."""
    L = _lib.lib()
    _need_gpu(edge_index)
    g = get_graph(edge_index, num_nodes)
    E = edge_index.shape[1]
    w = torch.empty(E, dtype=torch.float32, device=edge_index.device)
    _lib.check(L.sgs_er_weight(_ptr(g.edge_index), E, num_nodes, _ptr(g.out_ptr), _ptr(g.out_dst), int(walk_lengths), int(walks), int(seed),
                               _ptr(w), _stream()), "sgs_er_weight")
    return w if raw else torch.softmax(w * E ** -0.5, dim=0)


# ------------------------------------------------------------------ sparse node features (CitationFull-Cora: bag-of-words rows, 0.7 % dense)
class FeatCSR:
    """CSR of a sparse feature matrix x [N, F] and of its transpose, built once per graph: the node-level products of the first GCN
    layers, x W^T and d W = d Y^T x, are then two SpMMs over nnz(x) instead of two dense [N, F] x [F, H] GEMMs (model.py:159 feeds the
    raw bag-of-words rows to GCNConv.lin: 88 GFLOP per product at CitationFull-Cora's size, 0.6 GFLOP of it on non-zeros)."""
    __slots__ = ("N", "F", "nnz", "ptr", "col", "val", "tptr", "trow", "tval")


_FEAT_SPARSE_MAX_DENSITY = 0.05        # above this the library GEMM wins
_FEAT_SPARSE_MIN_ELEMS = 1 << 22       # small matrices: not worth a second code path


def feature_csr(x: torch.Tensor, build: bool = False):
    """FeatCSR of `x` if it is sparse enough (cached on the tensor, keyed by its version), else None.  Building reads the non-zero count
    back (set-up work, once per graph): only the models' FIRST layers ask for it (`build=True`, on the batch's resident x); the products
    themselves just look the cache up.  Inside a stream capture only an existing cache entry is used."""
    c = getattr(x, "_sgs_fcsr", None)
    if c is not None and c[1] == x._version:
        return c[0]
    if not build or not x.is_cuda or x.dim() != 2 or x.dtype != torch.float32 or x.numel() < _FEAT_SPARSE_MIN_ELEMS or x.requires_grad:
        return None
    if torch.cuda.is_current_stream_capturing():
        return None
    N, F_ = x.shape
    nz = x != 0
    nnz = int(nz.sum())
    fc = None
    if nnz <= _FEAT_SPARSE_MAX_DENSITY * x.numel() and nnz < 2**31:
        fc = FeatCSR()
        idx = torch.nonzero(nz)                                    # row-major: sorted by (row, col)
        rows, cols = idx[:, 0], idx[:, 1]
        val = x[rows, cols].contiguous()
        i32 = dict(dtype=torch.int32, device=x.device)
        fc.N, fc.F, fc.nnz = N, F_, nnz
        fc.ptr = torch.zeros(N + 1, **i32)
        fc.ptr[1:] = torch.cumsum(torch.bincount(rows, minlength=N), 0).to(torch.int32)
        fc.col, fc.val = cols.to(torch.int32).contiguous(), val
        order = torch.argsort(cols, stable=True)                   # the transpose: sorted by (col, row)
        fc.tptr = torch.zeros(F_ + 1, **i32)
        fc.tptr[1:] = torch.cumsum(torch.bincount(cols, minlength=F_), 0).to(torch.int32)
        fc.trow, fc.tval = rows[order].to(torch.int32).contiguous(), val[order].contiguous()
    try:
        x._sgs_fcsr = (fc, x._version)
    except Exception:
        pass
    return fc


def _x_wt(x, W):
    """x W^T: over the non-zeros of x when it has a FeatCSR (gathering rows of W^T), else the library GEMM."""
    fc = feature_csr(x)
    if fc is None:
        return x @ W.t()
    Wt = W.t().contiguous()                                        # [F, H]
    return _spmm(Wt, fc.ptr, fc.col, fc.val, None, None, ACT_NONE, 0.0, 0, 0, fc.N, W.shape[0], fc.nnz)


def _dyt_x(dY, x, W_shape):
    """d W [M, F] = d Y^T x: over the non-zeros of x^T when x has a FeatCSR, else sgs_gemm_tn."""
    L = _lib.lib()
    fc = feature_csr(x)
    M, Nn = W_shape
    if fc is not None:
        dWt = _spmm(dY.contiguous(), fc.tptr, fc.trow, fc.tval, None, None, ACT_NONE, 0.0, 0, 0, fc.F, M, fc.nnz)      # [F, M]
        return dWt.t().contiguous()
    K = x.shape[0]
    dW = torch.empty(M, Nn, dtype=torch.float32, device=x.device)
    ws = workspace(L.sgs_gemm_tn_workspace_bytes(K, M, Nn), x.device)
    _lib.check(L.sgs_gemm_tn(_ptr(dY, torch.float32), _ptr(x.contiguous(), torch.float32), K, M, Nn, _ptr(dW), ws.data_ptr(), ws.numel(), _stream()),
               "sgs_gemm_tn")
    return dW


# ------------------------------------------------------------------ one GCN layer as ONE autograd node
class _GCNLayer(torch.autograd.Function):
    """Y = act(A_hat (x W^T) + bias): the node-level product (library GEMM) and the propagation (K5) in a single
    autograd node -- the step is launch/host-bound at partition scale, so halving the Python nodes per layer matters.
    `xl` may be supplied (memoised x W^T shared by the learned and the random forward of one step)."""

    @staticmethod
    def forward(ctx, x, W, handle, bias, nm, act, p, seed, site, xl):
        gr = nm.graph
        if xl is None:
            xl = _x_wt(x, W)
        N, D = xl.shape
        Y = _spmm(xl, gr.in_ptr, gr.in_src, nm.what_in, nm.what_loop, bias, act, p, seed, site, N, D, gr.n_edges)
        ctx.nm, ctx.act, ctx.p = nm, act, p
        ctx.has_bias, ctx.has_handle = bias is not None, handle is not None
        ctx.save_for_backward(x, W, xl, Y if act != ACT_NONE else None)
        ctx.mark_non_differentiable(xl)
        ctx.set_materialize_grads(False)        # no zero-filled [N, D] gradient for the (never differentiated) xl output
        return Y, xl

    @staticmethod
    def backward(ctx, dY, _dxl_unused):
        L = _lib.lib()
        nm, gr = ctx.nm, ctx.nm.graph
        x, W, xl, Y = ctx.saved_tensors
        N, D = xl.shape
        dY = dY.contiguous()
        dx = dW = g = dbias = None
        want_db = ctx.has_bias and ctx.needs_input_grad[3]
        if ctx.act != ACT_NONE and want_db:
            dZ, dbias = _act_bwd_colsum(dY, Y, ctx.act, ctx.p)
        elif ctx.act != ACT_NONE:
            dZ = torch.empty_like(dY)
            _lib.check(L.sgs_act_bwd(_ptr(dY), _ptr(Y), dY.numel(), ctx.act, float(ctx.p), _ptr(dZ), _stream()), "sgs_act_bwd")
        else:
            dZ = dY
        need_x, need_W = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if need_x or need_W:
            dxl = _spmm(dZ, gr.out_ptr, gr.out_dst, nm.what_out, nm.what_loop, None, ACT_NONE, 0.0, 0, 0, N, D, gr.n_edges)
            if need_W:
                dW = _dyt_x(dxl, x, W.shape)
            if need_x:
                dx = dxl @ W
        if ctx.has_handle and ctx.needs_input_grad[2]:
            g = torch.empty(gr.n_edges + gr.N, dtype=torch.float32, device=dY.device)
            gw, gl = g[:gr.n_edges], g[gr.n_edges:]
            _lib.check(L.sgs_sddmm_csr(_ptr(dZ), _ptr(xl), N, D, gr.n_edges, _ptr(gr.in_ptr), _ptr(gr.in_src), _ptr(gr.in_eid),
                                       gw.data_ptr(), gl.data_ptr(), _stream()), "sgs_sddmm_csr")
        if want_db and dbias is None:
            dbias = _colsum(dZ)
        return dx, dW, (_handle_grad(nm, g) if g is not None else None), dbias, None, None, None, None, None, None


def gcn_layer(x, W, bias, nm: Norm, act=ACT_NONE, p=0.0, seed=0, site=0, xl=None):
    """act(A_hat (x W^T) + bias) with autograd to x, W, bias and (through nm.handle) the edge weights.
    Returns (Y, xl) where xl = x W^T (detached) can be passed back in for another graph over the same x, W."""
    _need_gpu(x, W, bias)
    if x.dtype != torch.float32 or x.dim() != 2 or x.shape[0] != nm.graph.N:
        raise RuntimeError("gcn_layer: x must be float32 [N, F]")
    return _GCNLayer.apply(x, W, nm.handle, bias, nm, act, float(p), int(seed), int(site), xl)
