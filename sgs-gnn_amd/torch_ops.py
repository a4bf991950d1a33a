"""`torch.ops.sgs.*`: the hot-path kernels registered as PyTorch custom operators (torch.library), the operator surface SURVEY.md
section 8b sketches -- tensor-only signatures, fake (meta) kernels for shape inference / tracing, autograd formulas registered on the
operators.  They are a second binding of the SAME C ABI calls (include/sgs_hip.h through sgs_gnn_amd.ops); the drop-in classes
(model.py, scorer.py, training.py) use sgs_gnn_amd.ops directly because they share CSR builds / normalisations across layers
through Python objects, which an operator signature cannot carry.

    torch.ops.sgs.sample_topq(p, prior, c, q, istest, noise, seed, offset, edge_index) -> (mask, eid, sampled_edge_index)
        sampling.py:91-96,134-139 + training_hybrid.py:83 (mode LEARNED; prior = None <=> istest)
    torch.ops.sgs.edge_score(node_codes, edge_index, W1, b1, w2, b2, p_drop, seed, offset, training) -> p [E]
        model.py:29-34 / 115-122 `_edge_score`; differentiable wrt node_codes, W1, b1, w2, b2
    torch.ops.sgs.gcn_propagate(x, edge_index, edge_weight, bias) -> act-free GCNConv propagate: A_hat(edge_weight) x + bias
        PyG gcn_norm + propagate (model.py:159-161 with x already transformed); differentiable wrt x, edge_weight, bias
    torch.ops.sgs.gat_propagate(xl, a_src, a_dst, edge_index, bias, negative_slope) -> sum_k alpha_k xl[src_k] + alpha_loop xl[i] + bias
        PyG GATConv (heads = 1) attention + aggregation (model.py:201-208); differentiable wrt xl, a_src, a_dst, bias

No CPU kernels are registered: a CPU tensor raises, as everywhere in this package.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import ops

SITE_SCORE = 2          # dropout site of the scorer's hidden layer (model.py:121), as sgs_gnn_amd.model.SITE_SCORE


# ------------------------------------------------------------------ sgs::sample_topq
@torch.library.custom_op("sgs::sample_topq", mutates_args=())
def sample_topq(p: Tensor, prior: Optional[Tensor], c: float, q: int, istest: bool, noise: Optional[Tensor], seed: int, offset: int,
                edge_index: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    r = ops.sample_topq(ops.SAMPLE_LEARNED, p.contiguous(), None if istest else prior, c, q, edge_index.contiguous(), noise=noise, seed=seed,
                        stream_id=offset, want_p=False)
    return r.mask, r.eid, r.edge_index


@sample_topq.register_fake
def _(p, prior, c, q, istest, noise, seed, offset, edge_index):
    E = p.shape[0]
    return (torch.empty(E, dtype=torch.bool, device=p.device), torch.empty(q, dtype=torch.int64, device=p.device),
            torch.empty(2, q, dtype=torch.int64, device=p.device))


# ------------------------------------------------------------------ sgs::edge_score (+ backward)
@torch.library.custom_op("sgs::edge_score", mutates_args=())
def edge_score(node_codes: Tensor, edge_index: Tensor, W1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor, p_drop: float, seed: int, offset: int,
               training: bool) -> Tensor:
    p = p_drop if training else 0.0
    with torch.no_grad():
        return ops.edge_score(node_codes, W1, b1, w2.reshape(1, -1), b2, edge_index, p=p, seed=seed, site=SITE_SCORE, edge_id_offset=offset)


@edge_score.register_fake
def _(node_codes, edge_index, W1, b1, w2, b2, p_drop, seed, offset, training):
    return torch.empty(edge_index.shape[1], dtype=torch.float32, device=node_codes.device)


@torch.library.custom_op("sgs::edge_score_backward", mutates_args=())
def edge_score_backward(grad_p: Tensor, node_codes: Tensor, edge_index: Tensor, W1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor, p_drop: float,
                        seed: int, offset: int) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    H = node_codes.shape[1]
    codes = node_codes.contiguous()
    ctx = SimpleNamespace(saved_tensors=(codes, torch.mm(codes, W1[:, H:].t()), W1.contiguous(), b1.contiguous(), w2.reshape(-1).contiguous(),
                                         b2.contiguous(), edge_index.contiguous()),
                          active=None, p=float(p_drop), seed=int(seed), site=SITE_SCORE, offset=int(offset), needs_input_grad=(True,) * 12)
    with torch.no_grad():
        g = ops._EdgeScore.backward(ctx, grad_p.contiguous())
    return g[0], g[1], g[2], g[3].reshape(w2.shape), g[4]


@edge_score_backward.register_fake
def _(grad_p, node_codes, edge_index, W1, b1, w2, b2, p_drop, seed, offset):
    return (torch.empty_like(node_codes), torch.empty_like(W1), torch.empty_like(b1), torch.empty_like(w2), torch.empty_like(b2))


def _edge_score_setup(ctx, inputs, output):
    node_codes, edge_index, W1, b1, w2, b2, p_drop, seed, offset, training = inputs
    ctx.save_for_backward(node_codes, edge_index, W1, b1, w2, b2)
    ctx.p, ctx.seed, ctx.offset = (p_drop if training else 0.0), seed, offset


def _edge_score_bwd(ctx, grad_p):
    node_codes, edge_index, W1, b1, w2, b2 = ctx.saved_tensors
    g = torch.ops.sgs.edge_score_backward(grad_p, node_codes, edge_index, W1, b1, w2, b2, ctx.p, ctx.seed, ctx.offset)
    return g[0], None, g[1], g[2], g[3], g[4], None, None, None, None


edge_score.register_autograd(_edge_score_bwd, setup_context=_edge_score_setup)


# ------------------------------------------------------------------ sgs::gcn_propagate (+ backward)
def _gcn_forward(x, edge_index, edge_weight, bias):
    graph = ops.get_graph(edge_index, x.shape[0])
    nm = ops.gcn_norm(graph, None if edge_weight is None else edge_weight)
    return nm, ops._spmm(x.contiguous(), graph.in_ptr, graph.in_src, nm.what_in, nm.what_loop, bias, ops.ACT_NONE, 0.0, 0, 0, x.shape[0], x.shape[1],
                         graph.n_edges)


@torch.library.custom_op("sgs::gcn_propagate", mutates_args=())
def gcn_propagate(x: Tensor, edge_index: Tensor, edge_weight: Optional[Tensor], bias: Optional[Tensor]) -> Tensor:
    with torch.no_grad():
        return _gcn_forward(x, edge_index.contiguous(), None if edge_weight is None else edge_weight.detach(), bias)[1]


@gcn_propagate.register_fake
def _(x, edge_index, edge_weight, bias):
    return torch.empty_like(x)


@torch.library.custom_op("sgs::gcn_propagate_backward", mutates_args=())
def gcn_propagate_backward(grad_y: Tensor, x: Tensor, edge_index: Tensor, edge_weight: Optional[Tensor], has_bias: bool) -> Tuple[Tensor, Tensor, Tensor]:
    """-> (dx, d edge_weight [n_edges] (zeros when edge_weight is None), d bias [D] (zeros when no bias))."""
    ei = edge_index.contiguous()
    with torch.enable_grad():
        xl = x.detach().requires_grad_(True)
        w = None if edge_weight is None else edge_weight.detach().requires_grad_(True)
        b = torch.zeros(x.shape[1], dtype=torch.float32, device=x.device, requires_grad=True) if has_bias else None
        nm = ops.gcn_norm(ops.get_graph(ei, x.shape[0]), w)
        y = ops.gcn_propagate(xl, nm, b)
        leaves = [t for t in (xl, w, b) if t is not None]
        gs = list(torch.autograd.grad(y, leaves, grad_y.contiguous(), allow_unused=True))
    dx = gs.pop(0)
    dw = gs.pop(0) if w is not None else torch.zeros(ei.shape[1], dtype=torch.float32, device=x.device)
    db = gs.pop(0) if b is not None else torch.zeros(x.shape[1], dtype=torch.float32, device=x.device)
    return dx, dw, db


@gcn_propagate_backward.register_fake
def _(grad_y, x, edge_index, edge_weight, has_bias):
    return (torch.empty_like(x), torch.empty(edge_index.shape[1], dtype=torch.float32, device=x.device),
            torch.empty(x.shape[1], dtype=torch.float32, device=x.device))


def _gcn_setup(ctx, inputs, output):
    x, edge_index, edge_weight, bias = inputs
    ctx.save_for_backward(x, edge_index, edge_weight)
    ctx.has_w, ctx.has_bias = edge_weight is not None, bias is not None


def _gcn_bwd(ctx, grad_y):
    x, edge_index, edge_weight = ctx.saved_tensors
    dx, dw, db = torch.ops.sgs.gcn_propagate_backward(grad_y, x, edge_index, edge_weight, ctx.has_bias)
    return dx, None, (dw if ctx.has_w else None), (db if ctx.has_bias else None)


gcn_propagate.register_autograd(_gcn_bwd, setup_context=_gcn_setup)


# ------------------------------------------------------------------ sgs::gat_propagate (+ backward)
@torch.library.custom_op("sgs::gat_propagate", mutates_args=())
def gat_propagate(xl: Tensor, a_src: Tensor, a_dst: Tensor, edge_index: Tensor, bias: Optional[Tensor], negative_slope: float) -> Tensor:
    with torch.no_grad():
        graph = ops.get_graph(edge_index.contiguous(), xl.shape[0])
        return ops.gat_aggregate(xl, a_src, a_dst, bias, graph, negative_slope)


@gat_propagate.register_fake
def _(xl, a_src, a_dst, edge_index, bias, negative_slope):
    return torch.empty_like(xl)


@torch.library.custom_op("sgs::gat_propagate_backward", mutates_args=())
def gat_propagate_backward(grad_y: Tensor, xl: Tensor, a_src: Tensor, a_dst: Tensor, edge_index: Tensor, has_bias: bool,
                           negative_slope: float) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    with torch.enable_grad():
        leaves = [t.detach().requires_grad_(True) for t in (xl, a_src, a_dst)]
        b = torch.zeros(xl.shape[1], dtype=torch.float32, device=xl.device, requires_grad=True) if has_bias else None
        graph = ops.get_graph(edge_index.contiguous(), xl.shape[0])
        y = ops.gat_aggregate(leaves[0], leaves[1], leaves[2], b, graph, negative_slope)
        gs = torch.autograd.grad(y, leaves + ([b] if b is not None else []), grad_y.contiguous())
    db = gs[3] if has_bias else torch.zeros(xl.shape[1], dtype=torch.float32, device=xl.device)
    return gs[0], gs[1], gs[2], db


@gat_propagate_backward.register_fake
def _(grad_y, xl, a_src, a_dst, edge_index, has_bias, negative_slope):
    return (torch.empty_like(xl), torch.empty_like(a_src), torch.empty_like(a_dst), torch.empty(xl.shape[1], dtype=torch.float32, device=xl.device))


def _gat_setup(ctx, inputs, output):
    xl, a_src, a_dst, edge_index, bias, negative_slope = inputs
    ctx.save_for_backward(xl, a_src, a_dst, edge_index)
    ctx.has_bias, ctx.slope = bias is not None, negative_slope


def _gat_bwd(ctx, grad_y):
    xl, a_src, a_dst, edge_index = ctx.saved_tensors
    g = torch.ops.sgs.gat_propagate_backward(grad_y, xl, a_src, a_dst, edge_index, ctx.has_bias, ctx.slope)
    return g[0], g[1], g[2], None, (g[3] if ctx.has_bias else None), None


gat_propagate.register_autograd(_gat_bwd, setup_context=_gat_setup)
