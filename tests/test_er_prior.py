"""Effective-resistance prior: the oracle's Monte-Carlo restatement approaches its exact expectation (CPU), and the HIP
kernel does too (GPU).  The reference's estimator is stochastic (python random.choice), hence 'parity unpinned'."""
import pytest
import torch

from oracle import sgs_oracle as O


def _graph(n=40, e=260, seed=2):
    import sgs_gnn_amd as S
    return S.synthetic_graph(n, e, 4, 3, seed=seed)


def test_oracle_monte_carlo_approaches_expectation():
    b = _graph(n=14, e=60)
    n = b.x.shape[0]
    exact = O.er_weight_expected(b.edge_index, n)
    mc = O.er_weight_monte_carlo(b.edge_index, n, r=1500, generator=torch.Generator().manual_seed(1), first=12)
    assert float((mc.double() - exact[:12]).abs().max()) < 0.05
    assert float(exact.min()) >= 0.0 and float(exact.max()) > 0.05


@pytest.mark.gpu
def test_hip_er_weights_match_expectation_and_are_seeded():
    import sgs_gnn_amd as S
    b = _graph()
    n = b.x.shape[0]
    ei = b.edge_index.to("cuda:0")
    exact = O.er_weight_expected(b.edge_index, n)
    w = S.ops.er_prior(ei, n, seed=7, walks=20000, raw=True).cpu().double()
    assert float((w - exact).abs().max()) < 0.02
    # reference setting (l = 4, r = 100): unbiased but noisy; seeded and reproducible
    w100a = S.ops.er_prior(ei, n, seed=7, raw=True)
    w100b = S.ops.er_prior(ei, n, seed=7, raw=True)
    w100c = S.ops.er_prior(ei, n, seed=8, raw=True)
    assert torch.equal(w100a, w100b) and not torch.equal(w100a, w100c)
    assert float((w100a.cpu().double() - exact).abs().mean()) < 0.05
    p = S.ops.er_prior(ei, n, seed=7)
    assert abs(float(p.sum()) - 1.0) < 1e-5 and p.shape == (ei.shape[1],)
    torch.testing.assert_close(p, torch.softmax(w100a * ei.shape[1] ** -0.5, dim=0))
