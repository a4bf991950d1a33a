"""GPU parity, end to end: the drop-in train() (hybrid / straight_through / two_pass, GCN and MLP
scorers) replayed on the golden fixtures captured from the REFERENCE's own training loop
(tests/golden/pipeline_*.pt): same initial state_dict, same Exp(1) noise for both draws."""
import argparse

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from conftest import load_golden
from oracle import sgs_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _args(fx):
    return argparse.Namespace(
        device=DEV, mode="learned", pipeline=fx["pipeline"], edge_mlp_type=fx["scorer"], conditional=fx["conditional"],
        sparse_edge_mlp=False, t_init=0.7, t_min=0.5, degree_bias_coef=0.3, reg1=True, reg2=True, regularizer1_coef=1.0,
        consist_reg_coef=0.5, hybrid_checkpoint=False, drop_rate=fx["drop"], lr=1e-3)


def _setup(fx):
    import sgs_gnn_amd as S
    Fin, H, C = fx["x"].shape[1], 16, 5
    m = S.GNNModel(Fin, H, C, dropout_prob=fx["drop"], edge_mlp_type=fx["scorer"])
    m.load_state_dict(fx["state0"])                     # same 12 / 10 keys as the reference
    m = m.to(DEV)
    opt_gnn = torch.optim.Adam([p for n, p in m.named_parameters() if "gcn" in n], lr=1e-3)             # main.py:100
    opt_edge = torch.optim.Adam([p for n, p in m.named_parameters() if "edge_prob_mlp" in n], lr=1e-3)  # main.py:122
    opt_all = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=5e-4)                               # main.py:123
    b = S.Batch(x=fx["x"], edge_index=fx["edge_index"], y=fx["y"], train_mask=fx["train_mask"], prob=fx["prob"]).to(DEV)
    return S, m, opt_gnn, opt_edge, opt_all, b


@pytest.mark.parametrize("name", ["hybrid_gcn", "st_gcn", "twopass_gcn", "hybrid_mlp", "twopass_mlp"])
def test_train_replays_reference_fixture(name):
    fx = load_golden(f"pipeline_{name}.pt")
    S, m, opt_gnn, opt_edge, opt_all, b = _setup(fx)
    args = _args(fx)
    crit = nn.CrossEntropyLoss()
    for epoch, st in enumerate(fx["steps"]):
        noise = list(st["noise"])
        args._sgs_noise = {}
        if fx["conditional"]:
            args._sgs_noise["prior"] = noise.pop(0).to(DEV)
        args._sgs_noise["sample"] = noise.pop(0).to(DEV)
        args._sgs_trace = tr = {}
        ret = S.train(args, epoch, 10, m, opt_gnn, opt_edge, opt_all, crit, [b], q=fx["q"], alternate_frequency=0)

        # --- draws: bit-exact edge sets
        if fx["conditional"]:
            ref_r = torch.zeros(fx["edge_index"].shape[1], dtype=torch.bool)
            ref_r[st["noise_idx"][0]] = True
            assert torch.equal(tr["rsei"].cpu(), fx["edge_index"][:, ref_r])
        torch.testing.assert_close(tr["edge_probs_full"].cpu(), st["scorer_out"][0].squeeze(), rtol=0, atol=2e-6)
        assert torch.equal(tr["sample"].mask.cpu(), st["mask"])
        assert torch.equal(tr["sample"].edge_index.cpu(), st["gnn_edge_index"][0])
        # --- weights / logits within 1e-4 (north_star: logits within 1e-4 fp32)
        torch.testing.assert_close(tr["w"].cpu(), st["gnn_edge_weight"][0], rtol=0, atol=2e-6)
        torch.testing.assert_close(tr["learned_out"].cpu(), st["gnn_out"][0], rtol=1e-4, atol=1e-4)
        if fx["conditional"]:
            torch.testing.assert_close(tr["random_out"].cpu(), st["gnn_out"][1], rtol=1e-4, atol=1e-4)
        # --- gate, loss, return tuple
        assert int(tr["update_edge_mlp"]) == st["ret_cond"]
        assert abs(ret[0] - st["ret_loss"]) < 1e-4
        assert abs(ret[1] - st["ret_temperature"]) < 1e-12 and ret[2] == st["ret_cond"] and ret[3] == st["ret_total"]
        # --- gradients of all parameters
        for k, v in m.named_parameters():
            g = st["grads"][k]
            if g.numel() == 0:
                assert v.grad is None or float(v.grad.abs().max()) == 0.0, k
            else:
                got = v.grad.cpu() if v.grad is not None else torch.zeros_like(g)
                torch.testing.assert_close(got, g, rtol=2e-3, atol=2e-6, msg=lambda s: f"step {epoch} grad {k}: {s}")
        # --- parameters after the optimiser steps (incl. the double Adam update of edge_prob_mlp.gcn*)
        for k, v in m.state_dict().items():
            torch.testing.assert_close(v.cpu(), st["state_after"][k], rtol=1e-4, atol=3e-6,
                                       msg=lambda s: f"step {epoch} param {k}: {s}")


def test_losses_against_torch():
    import sgs_gnn_amd as S
    g = torch.Generator().manual_seed(0)
    N, C, q = 300, 41, 5000
    logits = torch.randn(N, C, generator=g)
    y = torch.randint(0, C, (N,), generator=g)
    tm = torch.rand(N, generator=g) < 0.6
    sei = torch.randint(0, N, (2, q), generator=g)
    w = torch.rand(q, generator=g) * 0.98 + 0.01
    for c1, c2 in [(1.0, 0.5), (0.0, 0.5), (1.0, 0.0)]:
        lo = logits.clone().double().requires_grad_(True)
        wo = w.clone().double().requires_grad_(True)
        ce = F.cross_entropy(lo[tm], y[tm])
        l2, nvalid, lsum = O.reg1_loss(wo, sei, y, tm)
        l3 = O.consistency_loss(wo, sei, lo)
        tot = ce + c1 * l2 + c2 * l3
        tot.backward()
        ld = logits.clone().to(DEV).requires_grad_(True)
        wd = w.clone().to(DEV).requires_grad_(True)
        ced = S.ops.masked_cross_entropy(ld, y.to(DEV), tm.to(DEV))
        reg, terms = S.ops.edge_regularizers(wd, ld, sei.to(DEV), y.to(DEV), tm.to(DEV), c1, c2)
        (ced + reg).backward()
        t = terms.cpu()
        assert abs(float(ced) - float(ce)) < 1e-5
        assert abs(float(t[0]) - float(l2)) < 1e-5 and abs(float(t[1]) - float(l3)) < 1e-6
        assert int(t[2]) == nvalid and int(t[3]) == int(lsum)
        torch.testing.assert_close(wd.grad.cpu().double(), wo.grad, rtol=1e-4, atol=1e-9)
        torch.testing.assert_close(ld.grad.cpu().double(), lo.grad, rtol=1e-4, atol=1e-8)
    # gate counts == sklearn micro-F1 numerators (utils.calculate_f1)
    cnt = S.ops.masked_correct(logits.to(DEV), y.to(DEV), tm.to(DEV)).tolist()
    assert cnt == [O.correct_count(logits, y, tm), int(tm.sum())]
    assert abs(S.calculate_f1(logits.to(DEV), y.to(DEV), tm.to(DEV)) - O.micro_f1(logits, y, tm)) < 1e-12


def test_reg1_disabled_when_label_sum_not_above_one():
    """training_hybrid.py:125-128: loss2 = 0 unless sum(valid_edge_labels) > 1."""
    import sgs_gnn_amd as S
    N, C = 6, 3
    logits = torch.randn(N, C)
    y = torch.tensor([0, 0, 1, 2, 1, 2])
    tm = torch.tensor([True, True, True, True, False, False])
    sei = torch.tensor([[0, 0, 2, 4], [1, 2, 3, 5]])      # valid: (0,1) same, (0,2) diff, (2,3) diff -> label sum 1
    w = torch.tensor([0.3, 0.6, 0.2, 0.9])
    _, terms = S.ops.edge_regularizers(w.to(DEV), logits.to(DEV), sei.to(DEV), y.to(DEV), tm.to(DEV), 1.0, 0.5)
    t = terms.cpu()
    assert float(t[0]) == 0.0 and int(t[2]) == 3 and int(t[3]) == 1
    l2, _, _ = O.reg1_loss(w, sei, y, tm)
    assert l2 == 0


def test_mlp_scorer_with_prior_draw_raises_like_the_reference():
    """EdgeProbMLP scores only the q random edges when a random subgraph
    exists, and the sampler then fails on the [q] vs [E] shape mismatch (sampling.py:95)."""
    import sgs_gnn_amd as S
    fx = load_golden("pipeline_hybrid_mlp.pt")
    S_, m, opt_gnn, opt_edge, opt_all, b = _setup(fx)
    args = _args(fx)
    args.conditional = True
    with pytest.raises(RuntimeError, match="must match the size"):
        S.train(args, 0, 10, m, opt_gnn, opt_edge, opt_all, nn.CrossEntropyLoss(), [b], q=fx["q"])
