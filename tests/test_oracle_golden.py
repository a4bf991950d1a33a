"""CPU: the oracle (oracle/sgs_oracle.py) against the golden vectors produced by running the
reference itself (tests/golden/gen_golden.py).  This is what pins the oracle."""
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden
from oracle import sgs_oracle as O

PIPE = ["hybrid_gcn", "st_gcn", "twopass_gcn", "hybrid_mlp", "hybrid_gcn_drop", "twopass_mlp", "hybrid_mlp_drop"]


def test_multinomial_is_exponential_race():
    # torch.multinomial(no replacement) == topk(s / Exp(1)) under a shared generator state,
    # the identity every sampler kernel in this repo rests on.
    g = torch.Generator().manual_seed(7)
    s = torch.rand(5000, generator=g)
    st = g.get_state()
    idx = torch.multinomial(s, 700, replacement=False, generator=g)
    g.set_state(st)
    noise = torch.empty_like(s).exponential_(1, generator=g)
    _, oidx = O.exp_race_topq(s, noise, 700)
    assert torch.equal(idx, oidx)


def test_sampler_learned_cases_bit_exact():
    fx = load_golden("sampler.pt")
    for c in fx["learned"]:
        samples, Z = O.sampler_keys(c["p"], c["prior"], 0.3, c["istest"])
        assert torch.equal(samples, c["samples"])           # bit-exact keys numerator
        assert torch.equal(Z, c["Z"])
        mask, w = O.gumbel_softmax_sampling(c["prior"], c["p"], c["q"], 0.3, c["istest"], c["noise"])
        assert torch.equal(mask, c["mask"])
        assert int(mask.sum()) == c["q"]
        assert torch.equal(w, c["w"])
        assert torch.equal(c["edge_index"][:, mask], c["sampled_edge_index"])
        _, idx = O.exp_race_topq(samples, c["noise"], c["q"])
        assert torch.equal(idx, c["idx"])                    # even the race order


def test_sampler_prior_cases_bit_exact():
    fx = load_golden("sampler.pt")
    for c in fx["prior"]:
        assert torch.equal(O.add_degree_prior(c["edge_index"], 200), c["prob"])
        idx = O.prior_draw(c["prob"], c["noise"], c["q"])
        assert torch.equal(idx, c["idx"])
        assert torch.equal(c["edge_index"][:, idx], c["rsei"])


def test_random_edge_sampling():
    fx = load_golden("sampler.pt")
    for c in fx["randperm"]:
        assert torch.equal(O.random_edge_sampling(c["edge_index"], c["q"], c["perm"]), c["out"])


def test_sampler_tie_rule_lowest_edge_id_wins():
    s = torch.tensor([0.5, 0.25, 0.5, 0.5, 0.125])
    noise = torch.ones(5)
    _, idx = O.exp_race_topq(s, noise, 2)
    assert idx.tolist() == [0, 2]


def _replay(fx, dtype=torch.float32):
    """Replay the fixture's steps with the oracle; yields (step_fixture, oracle_result, grads, P)."""
    P = {k: v.clone().to(dtype).requires_grad_(True) for k, v in fx["state0"].items()}
    cfg = O.StepConfig(pipeline=fx["pipeline"], scorer=fx["scorer"], q=fx["q"], conditional=fx["conditional"],
                       drop_rate=fx["drop"])
    batch = dict(x=fx["x"].to(dtype), edge_index=fx["edge_index"], y=fx["y"], train_mask=fx["train_mask"],
                 prob=fx["prob"])
    st_edge, st_gnn = {}, {}
    for st in fx["steps"]:
        nz = O.StepNoise()
        noise = list(st["noise"])
        if fx["conditional"]:
            nz.prior_noise = noise.pop(0)
        nz.sample_noise = noise.pop(0)
        keeps = list(st["drop_keep"])
        if fx["drop"] > 0:
            assert fx["pipeline"] == "hybrid"
            if fx["scorer"] == "GCN":
                nz.masks_pass1 = O.Masks(enc_hidden=keeps[0], score_hidden=keeps[1])
                nz.gnn_keep_learned, nz.gnn_keep_random = keeps[2], keeps[3]
            else:               # EdgeProbMLP (model.py:21-25, 32): x, y endpoint dropouts, hidden dropout; then the GNN's (conditional False)
                nz.masks_pass1 = O.Masks(mlp_x=keeps[0], mlp_y=keeps[1], score_hidden=keeps[2])
                nz.gnn_keep_learned = keeps[3]
        R = O.learned_step_forward(P, batch, cfg, nz)
        for p_ in P.values():
            p_.grad = None
        R["loss"].backward()
        grads = {k: v.grad for k, v in P.items()}
        yield st, R, grads, P
        # optimiser steps exactly as training_hybrid.py:135-141 with main.py:100,122's name filters
        gnn_params = {k: v for k, v in P.items() if "gcn" in k}
        edge_params = {k: v for k, v in P.items() if "edge_prob_mlp" in k}
        with torch.no_grad():
            if R["update_edge_mlp"]:
                O.adam_step(edge_params, grads, st_edge)
            O.adam_step(gnn_params, grads, st_gnn)


@pytest.mark.parametrize("name", PIPE)
def test_pipeline_replay_matches_reference(name):
    fx = load_golden(f"pipeline_{name}.pt")
    for st, R, grads, P in _replay(fx):
        assert torch.equal(R["mask"], st["mask"])
        so = st["scorer_out"][0].squeeze()
        torch.testing.assert_close(R["edge_probs_full"].detach(), so, rtol=0, atol=1e-6)
        assert torch.equal(R["sei"], st["gnn_edge_index"][0])
        torch.testing.assert_close(R["w"].detach(), st["gnn_edge_weight"][0], rtol=0, atol=1e-6)
        torch.testing.assert_close(R["learned_out"].detach(), st["gnn_out"][0], rtol=1e-5, atol=1e-5)
        if fx["conditional"]:
            torch.testing.assert_close(R["random_out"].detach(), st["gnn_out"][1], rtol=1e-5, atol=1e-5)
        assert int(R["update_edge_mlp"]) == st["ret_cond"]
        assert abs(float(R["loss"].detach()) - st["ret_loss"]) < 1e-5
        for k, g in st["grads"].items():
            if g.numel() == 0:
                assert grads[k] is None or float(grads[k].abs().max()) == 0.0, k
            else:
                og = grads[k] if grads[k] is not None else torch.zeros_like(g)
                torch.testing.assert_close(og, g, rtol=1e-4, atol=1e-6, msg=lambda m: f"{k}: {m}")
    # after the last yield the generator applied the optimiser steps for the last fixture step too
    for k, v in fx["steps"][-1]["state_after"].items():
        torch.testing.assert_close(P[k].detach(), v, rtol=1e-5, atol=2e-6, msg=lambda m: f"{k}: {m}")


def test_fullsize_fixture_replay_matches_reference():
    """The oracle at Reddit-partition size (n=1013, F=602, H=256, C=41, E=210 000, q=100 000) against the reference's own step
    (tests/golden/pipeline_hybrid_gcn_s3size.pt; inputs rebuilt from integer hashes, see conftest.load_golden_fullsize)."""
    from conftest import load_golden_fullsize
    fx = load_golden_fullsize("pipeline_hybrid_gcn_s3size.pt")
    st = fx["steps"][0]
    P = {k: v.clone().requires_grad_(True) for k, v in fx["state0"].items()}
    cfg = O.StepConfig(pipeline="hybrid", scorer="GCN", q=fx["q"], conditional=True, drop_rate=0.0)
    batch = dict(x=fx["x"], edge_index=fx["edge_index"], y=fx["y"], train_mask=fx["train_mask"], prob=fx["prob"])
    nz = O.StepNoise(prior_noise=st["noise"][0], sample_noise=st["noise"][1])
    R = O.learned_step_forward(P, batch, cfg, nz)
    R["loss"].backward()
    assert torch.equal(R["mask"], st["mask"])
    torch.testing.assert_close(R["edge_probs_full"].detach(), st["scorer_out"], rtol=0, atol=1e-6)
    torch.testing.assert_close(R["w"].detach(), st["w_sampled"], rtol=0, atol=1e-6)
    torch.testing.assert_close(R["learned_out"].detach(), st["gnn_out"][0], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(R["random_out"].detach(), st["gnn_out"][1], rtol=1e-5, atol=1e-5)
    assert int(R["update_edge_mlp"]) == st["ret_cond"] == 1
    assert abs(float(R["loss"].detach()) - st["ret_loss"]) < 1e-5
    for k, g in st["grads"].items():
        torch.testing.assert_close(P[k].grad, g, rtol=1e-3, atol=1e-6, msg=lambda m: f"{k}: {m}")


def test_gcn_conv_against_dense_fp64_formula():
    # Third-party layer (PyG 2.3.1 GCNConv): parity unpinned; cross-check the restatement against
    # an independent dense formula, with existing self-loops and duplicate edges present.
    g = torch.Generator().manual_seed(3)
    N, Fin, Fout, E = 17, 5, 4, 90
    ei = torch.randint(0, N, (2, E), generator=g)
    ei[:, 5] = torch.tensor([3, 3])          # existing self loop keeps its weight
    ei[:, 9] = torch.tensor([3, 3])          # ... last one wins
    w = torch.rand(E, generator=g, dtype=torch.float64)
    x = torch.randn(N, Fin, generator=g, dtype=torch.float64)
    W = torch.randn(Fout, Fin, generator=g, dtype=torch.float64)
    b = torch.randn(Fout, generator=g, dtype=torch.float64)
    torch.testing.assert_close(O.gcn_conv(x, ei, w, W, b), O.gcn_dense_reference(x, ei, w, W, b), rtol=1e-12, atol=1e-12)
    torch.testing.assert_close(O.gcn_conv(x, ei, None, W, b), O.gcn_dense_reference(x, ei, None, W, b), rtol=1e-12,
                               atol=1e-12)


def test_reg1_isin_equivalence():
    # training_hybrid.py:110-113 uses isin(src, nonzero(train_mask)); the oracle uses a mask gather.
    g = torch.Generator().manual_seed(5)
    tm = torch.rand(30, generator=g) < 0.4
    src = torch.randint(0, 30, (200,), generator=g)
    tr = torch.nonzero(tm).squeeze()
    assert torch.equal(torch.isin(src, tr), tm[src])
