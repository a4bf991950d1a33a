"""Mirror of the reference's evaluate.py: `evaluate` and `ensemble_evaluate` (same signatures and
return tuple).  The three kernels of the training path are reused forward-only; differences that do
not change results: the scorer runs ONCE per batch instead of once per ensemble draw (model.eval():
no dropout, identical inputs => identical probabilities, evaluate.py:84), logits are averaged with a
running sum, and the per-split correct counts stay on the device until the end of the loader.

Test hook: `args._sgs_noise_eval = [noise_0, noise_1, ...]` feeds explicit Exp(1) noise per draw.
"""
from __future__ import annotations

import torch

from . import ops
from .sampling import draw_learned, draw_prior, random_edge_sampling


def _one_draw(args, model, batch, q, mode, edge_probs, noise):
    if mode == 'learned':
        if batch.edge_index.shape[1] > q:
            smp = draw_learned(None, edge_probs, batch.edge_index, q, args.degree_bias_coef, istest=True, noise=noise)
            w = ops.st_weights(edge_probs, None, args.degree_bias_coef, smp.stats, smp.eid)     # sampling.py:137-155
            return model(batch, smp.edge_index, w)
        return model(batch, batch.edge_index)
    if mode == 'random':
        if batch.edge_index.shape[1] > q:
            return model(batch, random_edge_sampling(batch.edge_index, q=q))
        return model(batch, batch.edge_index)
    if mode == 'edge':
        if batch.edge_index.shape[1] > q:
            return model(batch, draw_prior(batch.prob, batch.edge_index, q, noise=noise).edge_index)
        return model(batch, batch.edge_index)
    if mode == 'full':
        return model(batch, batch.edge_index)
    raise ValueError("Invalid mode. Choose 'learned', 'random', or 'full'.")


def _run(args, model, cluster_loader, device, q, mode, n_draws):
    model.eval()
    counts = None
    noises = list(getattr(args, "_sgs_noise_eval", None) or [])
    with torch.no_grad():
        for batch in cluster_loader:
            batch = batch.to(device)
            ops.new_memo_scope()
            edge_probs = None
            if mode == 'learned' and batch.edge_index.shape[1] > q:
                ops.get_pairs(batch.edge_index, batch.x.shape[0], build=True)     # once per partition (cached)
                edge_probs = model.edge_prob_mlp(batch.x, batch.edge_index).squeeze()         # encoder over the FULL batch graph
            out = None
            for _ in range(n_draws):
                o = _one_draw(args, model, batch, q, mode, edge_probs, noises.pop(0) if noises else None)
                out = o if out is None else out + o
            if n_draws > 1:
                out = out / n_draws                                                           # torch.mean(torch.stack(outs))
            c = torch.stack([ops.masked_correct(out, batch.y, m) for m in (batch.train_mask, batch.val_mask, batch.test_mask)])
            counts = c.to(torch.int64) if counts is None else counts + c
    if counts is None:
        return 0, 0, 0
    c = counts.tolist()
    # sum_b f1_b * n_b / sum_b n_b  ==  sum_b correct_b / sum_b n_b   (utils.calculate_f1 is accuracy)
    return tuple((c[s][0] / c[s][1]) if c[s][1] > 0 else 0 for s in range(3))


def evaluate(args, model, cluster_loader, device, q=500, mode=None, temperature=1.0):
    """evaluate.py:6-67."""
    return _run(args, model, cluster_loader, device, q, mode, 1)


def ensemble_evaluate(args, model, cluster_loader, device, q=500, mode=None, temperature=1.0):
    """evaluate.py:70-173: mean of the logits of args.num_samples_eval independent draws."""
    return _run(args, model, cluster_loader, device, q, mode, int(args.num_samples_eval))
