"""Edge scorers: mirror of model.py:8-145 (EdgeProbMLP, EdgeProbGCN, get_edge_mlp)."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .model import GCNConv, _DropoutClock, SITE_ENC, SITE_SCORE, SITE_MLP_X, SITE_MLP_Y
from .utils import segment


class EdgeProbGCN(nn.Module):
    """model.py:91-133: 2-layer GCN encoder over `random_sampled_edge_index` (or `edge_index`
    when None), then `_edge_score` for EVERY column of `edge_index`."""

    def __init__(self, in_channels, hidden_dim, dropout_prob=0.2):
        super().__init__()
        self.gcn1 = GCNConv(in_channels, hidden_dim)
        self.gcn2 = GCNConv(hidden_dim, hidden_dim)
        self.fc1 = nn.Linear(2 * hidden_dim, hidden_dim)
        self.dropout = nn.Dropout(dropout_prob)
        self.fc2 = nn.Linear(hidden_dim, 1)
        self.last_active = None

    def forward(self, node_features, edge_index, random_sampled_edge_index=None, use_checkpoint=False):
        # use_checkpoint (model.py:126-129) is accepted and moot: the fused scorer never stores
        # the [E,2H]/[E,H] activations and always recomputes them in backward.
        g = random_sampled_edge_index if random_sampled_edge_index is not None else edge_index
        N = node_features.shape[0]
        p = self.dropout.p if self.training else 0.0
        act = ops.ACT_RELU_DROPOUT if p > 0 else ops.ACT_RELU
        ops.feature_csr(node_features, build=True)                # sparse bag-of-words features: x W^T over their non-zeros (cached per graph)
        with segment(self, "edge_mlp_pre"):                       # model.py:103-112
            norm = ops.gcn_norm(ops.get_graph(g, N), None)
            out = self.gcn1(node_features, g, norm=norm, act=act, p=p, seed=_DropoutClock.next_seed(), site=SITE_ENC)
            out = self.gcn2(out, g, norm=norm, act=ops.ACT_RELU)
        self.last_active = ops.ActiveSet()
        with segment(self, "edge_score"):                         # model.py:124-131
            prob = ops.edge_score(out, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, edge_index,
                                  active=self.last_active, p=p, seed=_DropoutClock.next_seed(), site=SITE_SCORE)
        return prob.unsqueeze(1)


class EdgeProbMLP(nn.Module):
    """model.py:8-45: x = drop(relu(fcdim(X[src]))), y = ...(X[dst]) on the columns of
    `random_sampled_edge_index` (or `edge_index`), then `_edge_score(x, y)`.  The reference
    applies fcdim to the GATHERED [E',F] rows; relu(fcdim(.)) is row-wise, so it is hoisted to
    the N nodes (identical values); only the per-(edge, endpoint) dropout has to stay per edge."""

    def __init__(self, in_channels, hidden_dim, dropout_prob=0.2):
        super().__init__()
        self.dropout = nn.Dropout(dropout_prob)
        self.fcdim = nn.Linear(in_channels, hidden_dim)
        self.fc1 = nn.Linear(2 * hidden_dim, hidden_dim)
        self.fc2 = nn.Linear(hidden_dim, 1)
        self.last_active = None

    def forward(self, node_features, edge_index, random_sampled_edge_index=None, use_checkpoint=False):
        g = random_sampled_edge_index if random_sampled_edge_index is not None else edge_index
        A = F.relu(self.fcdim(node_features))
        p = self.dropout.p if self.training else 0.0
        self.last_active = ops.ActiveSet()
        if p == 0.0:
            prob = ops.edge_score(A, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, g,
                                  active=self.last_active)
        else:
            # per-(edge, endpoint) dropout (model.py:21-25) inside the scorer kernels: masks are hashed from (seed, site, edge, column),
            # nothing of size [E', H] is gathered, masked or concatenated (include/sgs_hip.h, "Endpoint-dropout scorer")
            sx, sy, ss = _DropoutClock.next_seed(), _DropoutClock.next_seed(), _DropoutClock.next_seed()
            if g is not edge_index:
                self.last_active = None        # scores of the RANDOM edges: the trainer's active set (ids of the batch graph) does not apply
            if A.shape[1] % 16 != 0:
                raise NotImplementedError("EdgeProbMLP with dropout needs hidden_dim % 16 == 0")
            prob = ops.edge_score_epd(A, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, g, active=self.last_active,
                                      p=p, seed=ss, site=SITE_SCORE, p_ep=p, seed_x=sx, site_x=SITE_MLP_X, seed_y=sy, site_y=SITE_MLP_Y)
        return prob.unsqueeze(1)


class SAGEConv(nn.Module):
    """PyG 2.3.1 SAGEConv(in, out) with its defaults (aggr='mean', root_weight=True, bias=True):
    out_i = lin_l(mean_{j -> i} x_j) + lin_r(x_i); keys `lin_l.weight`, `lin_l.bias`, `lin_r.weight`.
    The mean commutes with lin_l, so the aggregation runs on the H-wide transformed features."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.lin_l = nn.Linear(in_channels, out_channels, bias=True)
        self.lin_r = nn.Linear(in_channels, out_channels, bias=False)

    def forward(self, x, edge_index, *, act=ops.ACT_NONE, p=0.0, seed=0, site=0):
        nm = ops.mean_norm(ops.get_graph(edge_index, x.shape[0]))
        agg = ops.gcn_propagate(ops.linear_nobias(x, self.lin_l.weight), nm, None)
        out = agg + self.lin_l.bias + ops.linear_nobias(x, self.lin_r.weight)
        if act != ops.ACT_NONE:
            out = F.relu(out)
            if act == ops.ACT_RELU_DROPOUT and p > 0:
                keep = ops.dropout_keep(seed, site, out.shape[0], out.shape[1], p, out.device)
                out = out * keep / (1.0 - p)
        return out


class EdgeProbSAGE(nn.Module):
    """model.py:47-89 (`--edge_mlp_type GSAGE`): one SAGEConv encoder over `random_sampled_edge_index`
    (or `edge_index`), dropout(relu(.)), then `_edge_score` for every column of `edge_index`."""

    def __init__(self, in_channels, hidden_dim, dropout_prob=0.2):
        super().__init__()
        self.gcn1 = SAGEConv(in_channels, hidden_dim)
        self.fc1 = nn.Linear(2 * hidden_dim, hidden_dim)
        self.dropout = nn.Dropout(dropout_prob)
        self.fc2 = nn.Linear(hidden_dim, 1)
        self.last_active = None

    def forward(self, node_features, edge_index, random_sampled_edge_index=None, use_checkpoint=False):
        g = random_sampled_edge_index if random_sampled_edge_index is not None else edge_index
        p = self.dropout.p if self.training else 0.0
        act = ops.ACT_RELU_DROPOUT if p > 0 else ops.ACT_RELU
        with segment(self, "edge_mlp_pre"):                       # model.py:59-66
            out = self.gcn1(node_features, g, act=act, p=p, seed=_DropoutClock.next_seed(), site=SITE_ENC)
        self.last_active = ops.ActiveSet()
        with segment(self, "edge_score"):                         # model.py:80-87
            prob = ops.edge_score(out, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, edge_index,
                                  active=self.last_active, p=p, seed=_DropoutClock.next_seed(), site=SITE_SCORE)
        return prob.unsqueeze(1)


def get_edge_mlp(in_channels, hidden_dim, dropout_prob, edge_mlp_type='MLP'):
    """model.py:135-145."""
    if edge_mlp_type == 'MLP':
        return EdgeProbMLP(in_channels, hidden_dim, dropout_prob)
    if edge_mlp_type == 'GSAGE':
        return EdgeProbSAGE(in_channels, hidden_dim, dropout_prob)
    if edge_mlp_type == 'GCN':
        return EdgeProbGCN(in_channels, hidden_dim, dropout_prob)
    raise NotImplementedError
