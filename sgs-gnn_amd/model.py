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

    def _memo(self, x):
        """Memoised x W^T (plain tensor, no autograd graph) keyed on (x, W) identity + version: the hybrid step runs
        GNNModel twice on the same batch.x with the same weights (learned and random forward,
        training_hybrid.py:88,93); the weight gradient is computed by hand in the layer's backward."""
        W = self.lin.weight
        # `ops.memo_scope()`: the trainers / evaluators open a new scope per batch, so a memo never outlives the step it was
        # made in -- version counters alone are not enough (a replayed optimiser graph writes W without touching them)
        key = (ops.memo_scope(), x.data_ptr(), x._version, tuple(x.shape), W.data_ptr(), W._version)
        c = getattr(self, "_lin_cache", None)
        if c is not None and c[0] == key and c[1]() is x:
            return c[2], key
        return None, key

    def forward(self, x, edge_index, edge_weight=None, *, norm=None, act=ops.ACT_NONE, p=0.0, seed=0, site=0):
        if norm is None:
            norm = ops.gcn_norm(ops.get_graph(edge_index, x.shape[0]), edge_weight)
        xl, key = self._memo(x)
        y, xl = ops.gcn_layer(x, self.lin.weight, self.bias, norm, act=act, p=p, seed=seed, site=site, xl=xl)
        try:
            import weakref
            self._lin_cache = (key, weakref.ref(x), xl)
        except TypeError:
            self._lin_cache = None
        return y


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
        from .utils import segment
        x = data.x
        ops.feature_csr(x, build=True)                            # bag-of-words features: the first layer runs over their non-zeros (once per graph)
        with segment(self, "gnn_forward"):                        # model.py:156-163
            norm = ops.gcn_norm(ops.get_graph(edge_index, x.shape[0]), edge_weight)   # once for both layers
            p = self.dropout.p if self.training else 0.0
            act = ops.ACT_RELU_DROPOUT if p > 0 else ops.ACT_RELU
            h = self.gcn1(x, edge_index, norm=norm, act=act, p=p, seed=_DropoutClock.next_seed(), site=SITE_GNN)
            return self.gcn2(h, edge_index, norm=norm)


# ------------------------------------------------------------------ GAT head (model.py:189-208)
SITE_GAT_ATT, SITE_GAT_ACT = 16, 32          # attention dropout uses site, site + 1 per layer


class GATConv(nn.Module):
    """PyG 2.3.1 GATConv with heads = 1 as torch_geometric.nn.models.GAT instantiates it: parameters
    `lin_src.weight` (shared with `lin_dst.weight`), `att_src`, `att_dst` [1,1,out], `bias` [out]."""

    def __init__(self, in_channels, out_channels, heads=1, concat=True, negative_slope=0.2, dropout=0.0):
        super().__init__()
        if heads != 1:
            raise NotImplementedError("the reference's GATModel never passes `heads` on to GAT: heads = 1")
        self.in_channels, self.out_channels, self.negative_slope, self.dropout = in_channels, out_channels, negative_slope, dropout
        self.lin_src = nn.Linear(in_channels, out_channels, bias=False)
        self.lin_dst = self.lin_src
        self.att_src = nn.Parameter(torch.empty(1, 1, out_channels))
        self.att_dst = nn.Parameter(torch.empty(1, 1, out_channels))
        self.bias = nn.Parameter(torch.zeros(out_channels))
        a = math.sqrt(6.0 / (in_channels + out_channels))
        nn.init.uniform_(self.lin_src.weight, -a, a)
        b = math.sqrt(6.0 / (1 + out_channels))
        nn.init.uniform_(self.att_src, -b, b)
        nn.init.uniform_(self.att_dst, -b, b)

    def forward(self, x, edge_index, *, act=ops.ACT_NONE, p_act=0.0, seed=0, layer=0):
        graph = ops.get_graph(edge_index, x.shape[0])
        xl = self.lin_src(x)
        a_s, a_d = ops.gat_scores(xl, self.att_src, self.att_dst)      # node-level dots, one pass over x'
        p_att = self.dropout if self.training else 0.0
        return ops.gat_aggregate(xl, a_s, a_d, self.bias, graph, self.negative_slope, p_att, seed, SITE_GAT_ATT + 2 * layer, act,
                                 p_act, seed, SITE_GAT_ACT + layer)


class GAT(nn.Module):
    """torch_geometric.nn.models.GAT(in, hidden, num_layers=2, out_channels, dropout, act='relu')."""
    supports_edge_weight = False

    def __init__(self, in_channels, hidden_channels, num_layers, out_channels, dropout=0.0, act='relu'):
        super().__init__()
        if num_layers != 2 or act != 'relu':
            raise NotImplementedError
        self.dropout = dropout
        self.convs = nn.ModuleList([GATConv(in_channels, hidden_channels, dropout=dropout),
                                    GATConv(hidden_channels, out_channels, concat=False, dropout=dropout)])

    def forward(self, x, edge_index, edge_weight=None):
        # edge_weight is dropped, exactly as PyG's BasicGNN does for a conv without edge-weight support
        p = self.dropout if self.training else 0.0
        act = ops.ACT_RELU_DROPOUT if p > 0 else ops.ACT_RELU
        seed = _DropoutClock.next_seed()
        h = self.convs[0](x, edge_index, act=act, p_act=p, seed=seed, layer=0)
        return self.convs[1](h, edge_index, seed=seed, layer=1)


class GATModel(nn.Module):
    """model.py:189-208 (`heads` is accepted and unused, as in the reference)."""

    def __init__(self, in_channels, hidden_dim, num_classes, dropout_prob=0.3, heads=8, edge_mlp_type='MLP'):
        super().__init__()
        from .scorer import get_edge_mlp
        self.edge_prob_mlp = get_edge_mlp(in_channels, hidden_dim, dropout_prob, edge_mlp_type)
        self.dropout_prob = dropout_prob
        self.GAT = GAT(in_channels=in_channels, hidden_channels=hidden_dim, num_layers=2, out_channels=num_classes,
                       dropout=dropout_prob, act='relu')

    def forward(self, data, edge_index, edge_weight=None):
        from .utils import segment
        with segment(self, "gnn_forward"):
            return self.GAT(data.x, edge_index, edge_weight=edge_weight)


# ------------------------------------------------------------------ GIN head (model.py:165-184)
SITE_GIN = 48


class _MLP2(nn.Module):
    """torch_geometric.nn.MLP([a, b, b], act='relu', norm=None) as GIN.init_conv builds it (PyG 2.3.1, from memory):
    Linear -> ReLU -> Linear; state_dict keys `lins.0.{weight,bias}`, `lins.1.{weight,bias}`."""

    def __init__(self, a, b):
        super().__init__()
        self.lins = nn.ModuleList([nn.Linear(a, b), nn.Linear(b, b)])


class GINConv(nn.Module):
    """PyG GINConv(nn=MLP, eps=0, train_eps=False): out_i = nn((1 + eps) x_i + sum_{j -> i} x_j); `edge_weight` is not
    supported by GIN (BasicGNN calls conv(x, edge_index)).  The first Linear commutes with the sum, so the aggregation runs
    on the transformed features (hidden width instead of the input width): one SpMM with unit weights and diagonal 1 + eps."""

    def __init__(self, in_channels, out_channels, eps=0.0):
        super().__init__()
        self.nn = _MLP2(in_channels, out_channels)
        self.register_buffer("eps", torch.tensor([float(eps)]))      # state_dict key, as PyG (train_eps=False)
        self._eps = float(eps)                                       # host copy: reading the buffer would synchronise

    def forward(self, x, edge_index):
        nm = ops.sum_norm(ops.get_graph(edge_index, x.shape[0]), 1.0 + self._eps)
        l0, l1 = self.nn.lins
        h = ops.gcn_propagate(ops.linear_nobias(x, l0.weight), nm, l0.bias, ops.ACT_RELU)
        return ops.linear_nobias(h, l1.weight) + l1.bias


class GIN(nn.Module):
    """torch_geometric.nn.models.GIN(in, hidden, num_layers=2, out, dropout, act='relu'): conv -> relu -> dropout -> conv."""

    def __init__(self, in_channels, hidden_channels, num_layers, out_channels, dropout=0.0, act='relu'):
        super().__init__()
        if num_layers != 2 or act != 'relu':
            raise NotImplementedError("the reference instantiates GIN(num_layers=2, act='relu')")
        self.dropout = dropout
        self.convs = nn.ModuleList([GINConv(in_channels, hidden_channels), GINConv(hidden_channels, out_channels)])

    def forward(self, x, edge_index, edge_weight=None):
        h = F.relu(self.convs[0](x, edge_index))
        p = self.dropout if self.training else 0.0
        if p > 0:
            keep = ops.dropout_keep(_DropoutClock.next_seed(), SITE_GIN, h.shape[0], h.shape[1], p, h.device)
            h = h * keep / (1.0 - p)
        return self.convs[1](h, edge_index)


class GINModel(nn.Module):
    """model.py:165-184."""

    def __init__(self, in_channels, hidden_dim, num_classes, dropout_prob=0.3, edge_mlp_type='MLP'):
        super().__init__()
        from .scorer import get_edge_mlp
        self.edge_prob_mlp = get_edge_mlp(in_channels, hidden_dim, dropout_prob, edge_mlp_type)
        self.dropout_prob = dropout_prob
        self.GIN = GIN(in_channels=in_channels, hidden_channels=hidden_dim, num_layers=2, out_channels=num_classes,
                       dropout=dropout_prob, act='relu')

    def forward(self, data, edge_index, edge_weight=None):
        from .utils import segment
        with segment(self, "gnn_forward"):
            return self.GIN(data.x, edge_index, edge_weight=edge_weight)


# ------------------------------------------------------------------ Chebyshev head (model.py:211-230)
class ChebConv(nn.Module):
    """PyG ChebConv(in, out, K=1, normalization='sym') as the reference instantiates it.  With K = 1 only T_0(L) x = x is
    used: out = lins[0](x) + bias -- the Laplacian PyG normalises is never applied (so edge_index / edge_weight do not
    influence the output; kept in the signature).  Keys: `lins.0.weight`, `bias`."""

    def __init__(self, in_channels, out_channels, K=1, normalization='sym'):
        super().__init__()
        if K != 1:
            raise NotImplementedError("the reference instantiates ChebConv(K=1)")
        self.lins = nn.ModuleList([nn.Linear(in_channels, out_channels, bias=False)])
        self.bias = nn.Parameter(torch.zeros(out_channels))
        a = math.sqrt(6.0 / (in_channels + out_channels))            # glorot, as PyG's Linear(weight_initializer='glorot')
        nn.init.uniform_(self.lins[0].weight, -a, a)

    def forward(self, x, edge_index=None, edge_weight=None):
        return ops.linear_nobias(x, self.lins[0].weight) + self.bias


class ChebModel(nn.Module):
    """model.py:211-230."""

    def __init__(self, in_channels, hidden_dim, num_classes, dropout_prob=0.3, edge_mlp_type='MLP'):
        super().__init__()
        from .scorer import get_edge_mlp
        self.edge_prob_mlp = get_edge_mlp(in_channels, hidden_dim, dropout_prob, edge_mlp_type)
        self.dropout_prob = dropout_prob
        self.gcn1 = ChebConv(in_channels, hidden_dim, K=1, normalization='sym')
        self.dropout = nn.Dropout(dropout_prob)
        self.gcn2 = ChebConv(hidden_dim, num_classes, K=1, normalization='sym')

    def forward(self, data, edge_index, edge_weight=None):
        h = F.relu(self.gcn1(data.x, edge_index, edge_weight))
        p = self.dropout.p if self.training else 0.0
        if p > 0:
            keep = ops.dropout_keep(_DropoutClock.next_seed(), SITE_GNN, h.shape[0], h.shape[1], p, h.device)
            h = h * keep / (1.0 - p)
        return self.gcn2(h, edge_index, edge_weight)
