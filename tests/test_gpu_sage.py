"""GPU parity: GSAGE scorer (model.py:47-89; PyG SAGEConv: parity unpinned) and the device-side degree prior
(datasets.py:141-156)."""
import argparse

import pytest
import torch

from oracle import sgs_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_sageconv_and_scorer_vs_oracle():
    import sgs_gnn_amd as S
    g = torch.Generator().manual_seed(2)
    N, E, Fin, H = 120, 2500, 14, 32
    ei = torch.randint(0, N, (2, E), generator=g)
    ei[:, 4] = ei[0, 4]                       # an (i,i) edge counts as an ordinary in-edge
    ei[1, ei[1] == 7] = 8                     # node 7 has no in-edges: mean over the empty set = 0
    x = torch.randn(N, Fin, generator=g)
    m = S.GNNModel(Fin, H, 5, dropout_prob=0.0, edge_mlp_type="GSAGE")
    assert {"edge_prob_mlp.gcn1.lin_l.weight", "edge_prob_mlp.gcn1.lin_l.bias", "edge_prob_mlp.gcn1.lin_r.weight"} <= set(m.state_dict())
    P = {k: v.detach().clone().double().requires_grad_(True) for k, v in m.state_dict().items()}
    m = m.to(DEV)
    xd = x.clone().to(DEV).requires_grad_(True)
    pd = m.edge_prob_mlp(xd, ei.to(DEV)).squeeze()
    xo = x.clone().double().requires_grad_(True)
    po = O.edge_prob_sage(P, xo, ei).squeeze()
    assert float((pd.detach().cpu().double() - po.detach()).abs().max()) < 2e-6
    gp = torch.randn(E, generator=g)
    pd.backward(gp.to(DEV))
    po.backward(gp.double())

    def rel(a, r):
        return float((a.double().cpu() - r).abs().max()) / (float(r.abs().max()) + 1e-12)
    assert rel(xd.grad, xo.grad) < 1e-4
    for k in ("gcn1.lin_l.weight", "gcn1.lin_l.bias", "gcn1.lin_r.weight", "fc1.weight", "fc2.weight"):
        got = dict(m.edge_prob_mlp.named_parameters())[k].grad
        assert rel(got, P["edge_prob_mlp." + k].grad) < 1e-4, k


def test_degree_prior_on_device_matches_add_degree():
    import sgs_gnn_amd as S
    b = S.synthetic_graph(500, 40000, 4, 3, seed=5)
    want = O.add_degree_prior(b.edge_index, 500)
    got = S.ops.degree_prior(b.edge_index.to(DEV), 500).cpu()
    # the softmax normaliser is an fp32 sum of E terms in a different order: a uniform ~1e-5 relative factor
    torch.testing.assert_close(got, want, rtol=3e-5, atol=0)
    torch.testing.assert_close(got / got.sum(), want / want.sum(), rtol=2e-6, atol=0)
    assert abs(float(got.sum()) - 1.0) < 1e-4


def test_train_hybrid_with_gsage_scorer_runs_and_learns():
    import sgs_gnn_amd as S
    b = S.synthetic_graph(300, 6000, 12, 5, seed=1, train_frac=0.5).to(DEV)
    q = int(b.edge_index.shape[1] * 0.2)
    m = S.GNNModel(12, 32, 5, dropout_prob=0.3, edge_mlp_type="GSAGE").to(DEV)
    opt_gnn = torch.optim.Adam([p for n, p in m.named_parameters() if "gcn" in n], lr=1e-2)
    opt_edge = torch.optim.Adam([p for n, p in m.named_parameters() if "edge_prob_mlp" in n], lr=1e-2)
    opt_all = torch.optim.Adam(m.parameters(), lr=1e-2)
    args = argparse.Namespace(device=DEV, mode="learned", pipeline="hybrid", conditional=True, sparse_edge_mlp=True, t_init=0.7,
                              t_min=0.5, degree_bias_coef=0.3, reg1=True, reg2=True, regularizer1_coef=1.0, consist_reg_coef=0.5,
                              hybrid_checkpoint=False)
    S.fix_seeds(0)
    import contextlib, io

    def eval_ce():
        m.eval()
        with torch.no_grad():
            v = float(S.ops.masked_cross_entropy(m(b, b.edge_index), b.y, b.train_mask))
        m.train()
        return v
    before = eval_ce()
    losses = []
    with contextlib.redirect_stdout(io.StringIO()):
        for ep in range(30):
            losses.append(S.train(args, ep, 30, m, opt_gnn, opt_edge, opt_all, torch.nn.CrossEntropyLoss(), [b], q=q)[0])
    assert all(l == l for l in losses) and eval_ce() < before - 0.05
