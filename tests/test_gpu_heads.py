"""GPU: GIN and Cheb GNN heads (model.py:165-184, 211-230) against the oracle (fp64 autograd)."""
import pytest
import torch

from oracle import sgs_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _batch(S, n=70, e=600, f=9, seed=3):
    return S.synthetic_graph(n, e, f, 4, seed=seed)


@pytest.mark.parametrize("kind", ["GIN", "Cheb"])
def test_head_forward_backward_vs_oracle(kind):
    import sgs_gnn_amd as S
    torch.manual_seed(1)
    b = _batch(S)
    Model = S.GINModel if kind == "GIN" else S.ChebModel
    m = Model(9, 12, 4, dropout_prob=0.0, edge_mlp_type="GCN").to(DEV)
    # duplicate one edge: sums must count it twice
    ei = torch.cat([b.edge_index, b.edge_index[:, :5]], dim=1)
    bd = S.Batch(x=b.x.to(DEV), edge_index=ei.to(DEV), y=b.y.to(DEV), train_mask=b.train_mask.to(DEV))
    out = m(bd, bd.edge_index)
    P = {k: v.detach().cpu().double().requires_grad_(True) for k, v in m.state_dict().items() if v.dtype == torch.float32 and "eps" not in k}
    ref = O.gin_forward(P, b.x.double(), ei) if kind == "GIN" else O.cheb_forward(P, b.x.double())
    assert float((out.detach().cpu().double() - ref.detach()).abs().max()) < 1e-4
    g = torch.randn(out.shape, generator=torch.Generator().manual_seed(2))
    out.backward(g.to(DEV))
    ref.backward(g.double())
    for n_, p in m.named_parameters():
        if "edge_prob_mlp" in n_:
            assert p.grad is None
            continue
        r = P[n_].grad
        assert float((p.grad.cpu().double() - r).abs().max()) <= 2e-4 * (1.0 + float(r.abs().max())), n_


def test_gin_dropout_mask_and_training_loop():
    import argparse
    import sgs_gnn_amd as S
    torch.manual_seed(0)
    S.fix_seeds(0)
    bs = [S.synthetic_graph(150, e, 9, 4, seed=20 + i, device=DEV) for i, e in enumerate([5000, 900])]
    m = S.GINModel(9, 16, 4, dropout_prob=0.3, edge_mlp_type="GCN").to(DEV)
    og = S.FusedAdam([p for n, p in m.named_parameters() if "GIN" in n or "gcn" in n], lr=1e-2)       # main.py:100-109 routing
    oe = S.FusedAdam([p for n, p in m.named_parameters() if "edge_prob_mlp" in n], lr=1e-2)
    a = argparse.Namespace(device=DEV, mode="learned", pipeline="hybrid", conditional=True, sparse_edge_mlp=True, t_init=0.7, t_min=0.5,
                           degree_bias_coef=0.3, reg1=True, reg2=True, regularizer1_coef=1.0, consist_reg_coef=0.5, hybrid_checkpoint=False)
    for hip in (False, True):
        a.sgs_hipgraph = hip
        for ep in range(3):
            loss, _, cond, tot = S.train(a, ep, 3, m, og, oe, None, torch.nn.CrossEntropyLoss(), bs, q=1000)
            assert tot == 2 and loss == loss
    for n, p in m.named_parameters():
        assert torch.isfinite(p).all(), n
