"""Data-parallel glue for N > 1 GPUs (one process per GPU, torch.distributed: backend "nccl" is
RCCL over xGMI on ROCm, "gloo" on CPU for tests).

The METIS-partition stream shards by partition: rank r takes batches r, r+W, r+2W ... (weak
scaling, no data-path collective).  The one real exchange per step is the parameter-gradient
average: every gradient is packed into ONE flat fp32 bucket (~2.5 MB at H=256, F=602) together
with the gate outcome, summed with a single all-reduce and unpacked -- one collective per step
instead of one per tensor, sized for xGMI's per-link bandwidth.

Gate semantics under DP: each rank evaluates its own learned-vs-random gate on its own batch;
the scorer's optimiser steps on every rank iff at least one rank's gate chose "learned"
(ranks whose gate chose "random" contribute zero scorer gradients), so replicas stay identical.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def is_parallel() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def shard_batches(batches, rank: int, world: int):
    """Partition-level sharding: rank r owns batches r, r + world, ..."""
    return [b for i, b in enumerate(batches) if i % world == rank]


class GradSync:
    """Flat-bucket gradient averaging over the default process group."""

    def __init__(self, params):
        self.params = [p for p in params]
        self.numel = sum(p.numel() for p in self.params)
        self.flat = None

    def sync(self, learned_flag: bool) -> bool:
        """Average all .grad tensors across ranks (missing grads count as zeros).  Returns True iff
        any rank's gate chose "learned" this step."""
        if not is_parallel():
            return learned_flag
        p0 = self.params[0]
        if self.flat is None or self.flat.device != p0.device:
            self.flat = torch.zeros(self.numel + 1, dtype=torch.float32, device=p0.device)
        flat = self.flat
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                flat[off:off + n].zero_()
            else:
                flat[off:off + n].copy_(p.grad.reshape(-1))
            off += n
        flat[off] = 1.0 if learned_flag else 0.0
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        world = dist.get_world_size()
        any_learned = bool(flat[off].item() > 0.5)
        off = 0
        for p in self.params:
            n = p.numel()
            g = flat[off:off + n].view_as(p) / world
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += n
        return any_learned
