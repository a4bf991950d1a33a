"""GPU parity: GAT attention (K8) vs the oracle's restatement of PyG GATConv (heads = 1) and an
independent dense masked-softmax formula.  Third-party layer: parity unpinned."""
import pytest
import torch
import torch.nn.functional as F

from oracle import sgs_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _graph(N, E, seed):
    g = torch.Generator().manual_seed(seed)
    ei = torch.randint(0, N, (2, E), generator=g)
    if E > 8:
        ei[:, 1] = ei[0, 1]          # existing self loops are removed by GATConv
        ei[:, 5] = ei[0, 5]
    return ei, g


def dense_gat(x, ei, W, a_s, a_d, b, slope=0.2):
    """Independent dense formula (fp64): masked softmax over an [N, N] score matrix with multiplicities."""
    N = x.shape[0]
    xl = x @ W.t()
    s, d = xl @ a_s, xl @ a_d
    score = F.leaky_relu(d[:, None] + s[None, :], slope)          # [dst, src]
    cnt = torch.zeros(N, N, dtype=x.dtype)
    for e in range(ei.shape[1]):
        if ei[0, e] != ei[1, e]:
            cnt[ei[1, e], ei[0, e]] += 1
    cnt += torch.eye(N, dtype=x.dtype)
    w = cnt * torch.exp(score - score.max())
    alpha = w / w.sum(dim=1, keepdim=True)
    return alpha @ xl + b


@pytest.mark.parametrize("N,E,Fin,D", [(30, 200, 7, 16), (200, 5000, 20, 256), (64, 900, 9, 5), (25, 0, 4, 8)])
def test_gatconv_forward_backward(N, E, Fin, D):
    from sgs_gnn_amd.model import GATConv
    ei, g = _graph(N, E, N + D)
    x = torch.randn(N, Fin, generator=g)
    conv = GATConv(Fin, D)
    with torch.no_grad():
        conv.bias.uniform_(-0.3, 0.3)
    W, a_s, a_d, b = (t.detach().clone().double() for t in (conv.lin_src.weight, conv.att_src.reshape(-1), conv.att_dst.reshape(-1), conv.bias))
    gy = torch.randn(N, D, generator=g)
    leaves = [t.clone().requires_grad_(True) for t in (x.double(), W, a_s, a_d, b)]
    yo = O.gat_conv(leaves[0], ei, leaves[1], leaves[2], leaves[3], leaves[4])
    torch.testing.assert_close(yo.detach(), dense_gat(x.double(), ei, W, a_s, a_d, b), rtol=1e-10, atol=1e-10)
    yo.backward(gy.double())

    conv = conv.to(DEV)
    xd = x.clone().to(DEV).requires_grad_(True)
    yd = conv(xd, ei.to(DEV))
    yd.backward(gy.to(DEV))

    def rel(a, r):
        return float((a.double().cpu() - r).abs().max()) / (float(r.abs().max()) + 1e-12)
    assert rel(yd.detach(), yo.detach()) < 1e-5
    assert rel(xd.grad, leaves[0].grad) < 1e-4
    assert rel(conv.lin_src.weight.grad, leaves[1].grad) < 1e-4
    assert rel(conv.att_src.grad.reshape(-1), leaves[2].grad) < 1e-4
    assert rel(conv.att_dst.grad.reshape(-1), leaves[3].grad) < 1e-4
    assert rel(conv.bias.grad, leaves[4].grad) < 1e-4


def test_gat_model_state_dict_and_training_mode_dropout():
    import sgs_gnn_amd as S
    from sgs_gnn_amd import model as M
    m = S.GATModel(12, 16, 5, dropout_prob=0.3, edge_mlp_type="GCN")
    keys = set(m.state_dict())
    for i in (0, 1):
        for k in ("att_src", "att_dst", "bias", "lin_src.weight", "lin_dst.weight"):
            assert f"GAT.convs.{i}.{k}" in keys
    assert [n for n, _ in m.named_parameters() if "GAT" in n]           # main.py:107 name filter
    N, E = 80, 1200
    ei, g = _graph(N, E, 3)
    x = torch.randn(N, 12, generator=g)

    class D_:
        pass
    data = D_()
    data.x = x.to(DEV)
    m = m.to(DEV).train()
    p = 0.3
    M.set_dropout_seed(5)
    seed = (M._DropoutClock.base * 0x9E3779B97F4A7C15 + 1 * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
    out = m(data, ei.to(DEV), torch.rand(E, device=DEV))                 # edge_weight is ignored
    # replay with exported masks through the oracle
    keep_e0 = S.ops.dropout_keep(seed, M.SITE_GAT_ATT, E, 1, p, DEV).cpu().reshape(-1)
    keep_l0 = S.ops.dropout_keep(seed, M.SITE_GAT_ATT + 1, N, 1, p, DEV).cpu().reshape(-1)
    keep_e1 = S.ops.dropout_keep(seed, M.SITE_GAT_ATT + 2, E, 1, p, DEV).cpu().reshape(-1)
    keep_l1 = S.ops.dropout_keep(seed, M.SITE_GAT_ATT + 3, N, 1, p, DEV).cpu().reshape(-1)
    keep_h = S.ops.dropout_keep(seed, M.SITE_GAT_ACT, N, 16, p, DEV).cpu()
    nl = ei[0] != ei[1]
    c0, c1 = m.GAT.convs[0], m.GAT.convs[1]
    P = lambda t: t.detach().cpu().double()
    h = O.gat_conv(x.double(), ei, P(c0.lin_src.weight), P(c0.att_src).reshape(-1), P(c0.att_dst).reshape(-1), P(c0.bias),
                   att_mask=torch.cat([keep_e0[nl], keep_l0]).double(), p=p)
    h = F.relu(h) * keep_h.double() / (1 - p)
    o = O.gat_conv(h, ei, P(c1.lin_src.weight), P(c1.att_src).reshape(-1), P(c1.att_dst).reshape(-1), P(c1.bias),
                   att_mask=torch.cat([keep_e1[nl], keep_l1]).double(), p=p)
    assert float((out.detach().cpu().double() - o).abs().max()) / float(o.abs().max()) < 1e-5
    out.sum().backward()                                                # backward with dropout runs


def test_train_straight_through_with_gat_head():
    """Config 4 plumbing: --pipeline straight_through --GNN GAT; the GNN ignores edge weights, so the
    scorer receives gradient only through reg1 / reg2."""
    import argparse
    import sgs_gnn_amd as S
    b = S.synthetic_graph(300, 6000, 12, 5, seed=1, train_frac=0.5).to(DEV)
    q = int(b.edge_index.shape[1] * 0.2)
    torch.manual_seed(11)                                               # the initial weights must not depend on which tests ran before
    m = S.GATModel(12, 32, 5, dropout_prob=0.3, edge_mlp_type="GCN").to(DEV)
    opt_gnn = torch.optim.Adam([p for n, p in m.named_parameters() if "GAT" in n], lr=1e-2)            # main.py:107
    opt_edge = torch.optim.Adam([p for n, p in m.named_parameters() if "edge_prob_mlp" in n], lr=1e-2)
    opt_all = torch.optim.Adam(m.parameters(), lr=1e-2)
    args = argparse.Namespace(device=DEV, mode="learned", pipeline="straight_through", conditional=True, sparse_edge_mlp=False,
                              t_init=0.7, t_min=0.5, degree_bias_coef=0.3, reg1=True, reg2=True, regularizer1_coef=1.0,
                              consist_reg_coef=0.5)
    S.fix_seeds(0)

    def eval_ce():
        m.eval()
        with torch.no_grad():
            v = float(S.ops.masked_cross_entropy(m(b, b.edge_index), b.y, b.train_mask))
        m.train()
        return v
    before = eval_ce()
    losses = []
    for ep in range(30):
        ret = S.train(args, ep, 30, m, opt_gnn, opt_edge, opt_all, torch.nn.CrossEntropyLoss(), [b], q=q)
        losses.append(ret[0])
    assert all(l == l for l in losses)                                  # finite
    # the per-step loss mixes two objectives (CE vs CE + regularisers, by gate branch): judge learning by the
    # eval-mode full-graph cross-entropy on the train nodes instead
    after = eval_ce()
    print("gat_st_eval_ce", before, after)
    assert after < before - 0.03
