"""Mirror of the reference's sampling.py (gumbel_softmax_sampling, random_edge_sampling) on the
fused HIP sampler.  Randomness is counter-based: every draw consumes one (seed, stream_id)
pair from the process-wide noise clock (`manual_seed`), or explicit `noise` for parity tests."""
from __future__ import annotations

import torch

from . import ops


class _NoiseClock:
    seed = 42
    tick = 0

    @classmethod
    def next(cls):
        cls.tick += 1
        return cls.seed, cls.tick


def manual_seed(seed: int) -> None:
    """Seed the sampler's noise streams and the dropout masks (utils.fix_seeds counterpart)."""
    from .model import set_dropout_seed
    _NoiseClock.seed, _NoiseClock.tick = int(seed) & 0xFFFFFFFFFFFFFFFF, 0
    set_dropout_seed(seed)


def draw_learned(prior, edge_probs, edge_index, q, degree_bias_coef=0.3, istest=False, noise=None, want_p=False) -> ops.SampleResult:
    """K2+K3: the learned draw; `edge_probs` is used detached (sampling itself is not differentiable)."""
    E = edge_index.shape[1]
    if edge_probs.numel() != E or (not istest and prior.numel() != E):
        # the reference fails the same way (e.g. EdgeProbMLP scoring only the q random edges):
        raise RuntimeError(f"The size of tensor a ({edge_probs.numel()}) must match the size of tensor b "
                           f"({E if istest else prior.numel()}) at non-singleton dimension 0")
    seed, sid = (0, 0) if noise is not None else _NoiseClock.next()
    return ops.sample_topq(ops.SAMPLE_LEARNED, edge_probs.detach().contiguous(), None if istest else prior, degree_bias_coef, q,
                           edge_index, noise=noise, seed=seed, stream_id=sid, want_p=want_p)


def draw_prior(prob, edge_index, q, noise=None) -> ops.SampleResult:
    """K0: training_hybrid.py:46-48 (softmax(batch.prob) -> multinomial -> column gather)."""
    seed, sid = (0, 0) if noise is not None else _NoiseClock.next()
    return ops.sample_topq(ops.SAMPLE_PRIOR, prob, None, 0.0, q, edge_index, noise=noise, seed=seed, stream_id=sid, want_p=False)


def gumbel_softmax_sampling(batch, edge_probs, edge_index, q=500, temperature=1.0, degree_bias_coef=0.3, log=False,
                            istest=False, epoch=-1, *, noise=None):
    """sampling.py:91-155.  Returns (mask BoolTensor[E], weights FloatTensor[q] in original edge
    order, clamped to [0,1], autograd-connected to edge_probs).  `temperature`, `log`, `epoch`
    are dead in the reference and here.  The draw's details ride on `mask._sgs_sample`."""
    prior = None if istest else batch.prob
    r = draw_learned(prior, edge_probs, edge_index, q, degree_bias_coef, istest, noise)
    w = ops.st_weights(edge_probs, prior, degree_bias_coef, r.stats, r.eid)
    r.mask._sgs_sample = r
    return r.mask, w


def random_edge_sampling(edge_index, q, *, perm=None):
    """sampling.py:159-163: `edge_index[:, torch.randperm(E)[:q]]`, a uniformly random q-subset of the columns.
    On the device the subset is drawn by the sampler with uniform weights (an exponential race with equal weights IS a uniform
    draw without replacement; noise from the process-wide noise clock, `manual_seed`) and emitted in ORIGINAL edge order -- the
    reference's column order is the permutation's, which no consumer depends on (every GNN layer sums over incoming edges).
    `perm` (parity hook): an explicit permutation of range(E); the result is then exactly `edge_index[:, perm[:q]]`."""
    if perm is not None:
        return ops.gather_columns(edge_index.contiguous(), perm[:q].to(edge_index.device))
    seed, sid = _NoiseClock.next()
    return ops.sample_topq(ops.SAMPLE_LEARNED, None, None, 0.0, q, edge_index.contiguous(), seed=seed, stream_id=sid, want_p=False).edge_index
