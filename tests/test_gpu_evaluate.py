"""GPU parity: evaluate / ensemble_evaluate (evaluate.py:6-173) vs the oracle with explicit noise."""
import argparse

import pytest
import torch

from conftest import load_golden
from oracle import sgs_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_ensemble_evaluate_matches_oracle():
    import sgs_gnn_amd as S
    fx = load_golden("pipeline_hybrid_gcn.pt")
    m = S.GNNModel(fx["x"].shape[1], 16, 5, dropout_prob=0.3, edge_mlp_type="GCN")
    m.load_state_dict(fx["state0"])
    m = m.to(DEV)
    n = fx["x"].shape[0]
    g = torch.Generator().manual_seed(1)
    val = torch.rand(n, generator=g) < 0.5
    b = S.Batch(x=fx["x"], edge_index=fx["edge_index"], y=fx["y"], train_mask=fx["train_mask"], val_mask=val & ~fx["train_mask"],
                test_mask=~val & ~fx["train_mask"], prob=fx["prob"])
    E, q, draws = fx["edge_index"].shape[1], fx["q"], 5
    noises = [torch.empty(E).exponential_(1, generator=g) for _ in range(draws)]
    args = argparse.Namespace(degree_bias_coef=0.3, num_samples_eval=draws)
    args._sgs_noise_eval = [t.to(DEV) for t in noises]
    got = S.ensemble_evaluate(args, m, [b], DEV, q=q, mode="learned")

    P = fx["state0"]
    probs = O.edge_prob_gcn(P, fx["x"], fx["edge_index"], None).squeeze()          # eval: encoder over the full graph
    outs = []
    for nz in noises:
        mask, w = O.gumbel_softmax_sampling(None, probs, q, 0.3, True, nz)
        outs.append(O.gnn_forward(P, fx["x"], fx["edge_index"][:, mask], w))
    out = torch.stack(outs).mean(0)
    want = tuple(O.micro_f1(out, fx["y"], mk) for mk in (b.train_mask, b.val_mask, b.test_mask))
    assert got == pytest.approx(want, abs=1e-12)
    # single-draw evaluate and the other modes run and return fractions
    args._sgs_noise_eval = [noises[0].to(DEV)]
    one = S.evaluate(args, m, [b], DEV, q=q, mode="learned")
    mask, w = O.gumbel_softmax_sampling(None, probs, q, 0.3, True, noises[0])
    o1 = O.gnn_forward(P, fx["x"], fx["edge_index"][:, mask], w)
    assert one == pytest.approx(tuple(O.micro_f1(o1, fx["y"], mk) for mk in (b.train_mask, b.val_mask, b.test_mask)), abs=1e-12)
    full = S.evaluate(args, m, [b], DEV, q=q, mode="full")
    of = O.gnn_forward(P, fx["x"], fx["edge_index"], None)
    assert full == pytest.approx(tuple(O.micro_f1(of, fx["y"], mk) for mk in (b.train_mask, b.val_mask, b.test_mask)), abs=1e-12)
    for mode in ("random", "edge"):
        r = S.evaluate(args, m, [b], DEV, q=q, mode=mode)
        assert all(0.0 <= v <= 1.0 for v in r)
    with pytest.raises(ValueError):
        S.evaluate(args, m, [b], DEV, q=q, mode="bogus")
