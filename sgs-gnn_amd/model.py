"""Host-side mirror of the reference's model.py: same class names, constructor signatures,
forward signatures and state_dict keys, with every sparse / per-edge
op running in libsgs_hip.so.  Dense node-level X W^T products are library GEMMs (torch ->
hipBLASLt), as the hot-path scope allows.

Dropout: the reference draws nn.Dropout masks from torch's global generator; here masks are
counter-based, a pure function of (seed, site, row, col) (sgs_dropout_keep), fused into the
producing kernel.  `set_dropout_seed` / the per-forward step counter make every training
forward draw fresh masks.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops

# dropout sites (model.py:107 scorer encoder, :121 _edge_score hidden, :160 GNN hidden, :21-25 MLP pre)
SITE_ENC, SITE_SCORE, SITE_GNN, SITE_MLP_X, SITE_MLP_Y = 1, 2, 3, 4, 5


class _DropoutClock:
    """Process-wide source of dropout seeds: seed = hash(base_seed, forward counter)."""
    base = 0x5D5C0FFEE
    tick = 0

    @classmethod
    def next_seed(cls) -> int:
        cls.tick += 1
        return (cls.base * 0x9E3779B97F4A7C15 + cls.tick * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF


def set_dropout_seed(seed: int) -> None:
    _DropoutClock.base = int(seed) & 0xFFFFFFFFFFFFFFFF
    _DropoutClock.tick = 0


class GCNConv(nn.Module):
    """PyG GCNConv(in, out) as the reference instantiates it (model.py:94-95,151-153): keys
    `lin.weight` [out,in] (glorot), `bias` [out] (zeros).  forward = lin -> gcn_norm ->
    propagate -> + bias, with optional fused ReLU / dropout epilogue."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin = nn.Linear(in_channels, out_channels, bias=False)
        self.bias = nn.Parameter(torch.zeros(out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        a = math.sqrt(6.0 / (self.in_channels + self.out_channels))
        nn.init.uniform_(self.lin.weight, -a, a)
        nn.init.zeros_(self.bias)

    def forward(self, x, edge_index, edge_weight=None, *, norm=None, act=ops.ACT_NONE, p=0.0, seed=0, site=0):
        if norm is None:
            norm = ops.gcn_norm(ops.get_graph(edge_index, x.shape[0]), edge_weight)
        return ops.gcn_propagate(self.lin(x), norm, self.bias, act=act, p=p, seed=seed, site=site)


class GNNModel(nn.Module):
    """model.py:147-164."""

    def __init__(self, in_channels, hidden_dim, num_classes, dropout_prob=0.3, edge_mlp_type='MLP'):
        super().__init__()
        from .scorer import get_edge_mlp
        self.edge_prob_mlp = get_edge_mlp(in_channels, hidden_dim, dropout_prob, edge_mlp_type)
        self.gcn1 = GCNConv(in_channels, hidden_dim)
        self.dropout = nn.Dropout(dropout_prob)
        self.gcn2 = GCNConv(hidden_dim, num_classes)

    def forward(self, data, edge_index, edge_weight=None):
        x = data.x
        norm = ops.gcn_norm(ops.get_graph(edge_index, x.shape[0]), edge_weight)   # once for both layers
        p = self.dropout.p if self.training else 0.0
        act = ops.ACT_RELU_DROPOUT if p > 0 else ops.ACT_RELU
        h = self.gcn1(x, edge_index, norm=norm, act=act, p=p, seed=_DropoutClock.next_seed(), site=SITE_GNN)
        return self.gcn2(h, edge_index, norm=norm)
