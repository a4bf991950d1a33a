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
Opt-in `args.sgs_dp_global_gate` (graph mode with FusedAdam; what `bench.py --gpus N` sets): ONE gate per step over the union
of the ranks' batches -- the four counts are summed over ranks (16 bytes) and micro-F1 is compared on the sums -- so all ranks
take the same branch and no rank waits through another's learned backward.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def is_parallel() -> bool:
    """A process group with more than one rank -- or, with SGS_DP_FORCE=1, any initialised group: lets a ONE-GPU box drive the
    whole data-parallel path (bucket all-reduce between graph replays, gate sum, shared optimiser graph) through RCCL itself."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("SGS_DP_FORCE") == "1"


def shard_batches(batches, rank: int, world: int):
    """Partition-level sharding: rank r owns batches r, r + world, ...  Shards may differ in length by one (P % world != 0):
    `train` agrees on the longest shard at the start of every epoch and the shorter ranks finish with null steps."""
    return [b for i, b in enumerate(batches) if i % world == rank]


class GradSync:
    """Flat-bucket gradient averaging over the default process group.

    Per step: one `_foreach_copy_` packs the present gradients into views of ONE persistent fp32
    bucket (absent gradients are zeroed views), one all-reduce sums it, one scale averages it, and the
    parameters' `.grad` are pointed at the views (no unpack copies).  No host synchronisation."""

    def __init__(self, params):
        self.params = [p for p in params]
        self.sizes = [p.numel() for p in self.params]
        self.numel = sum(self.sizes)
        self.flat = None
        self.views = None

    def _ensure(self, device):
        if self.flat is None or self.flat.device != device:
            # one extra word rides behind the gradients: the number of ranks whose gate chose "learned" (graph mode: the
            # captured backward writes it, the same all-reduce sums it, the captured optimiser steps read it on the device)
            self.flat = torch.zeros(self.numel + 1, dtype=torch.float32, device=device)
            self.views = [v.view_as(p) for v, p in zip(self.flat[:self.numel].split(self.sizes), self.params)]
            self.flag = self.flat[self.numel:self.numel + 1]

    def max_steps(self, n_local: int, device) -> int:
        """Largest per-rank step count of this epoch (one small all-reduce per epoch)."""
        if not is_parallel():
            return n_local
        t = torch.tensor([n_local], dtype=torch.int64, device=device if dist.get_backend() != "gloo" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return int(t.item())

    def all_reduce_bucket(self) -> None:
        """Graph mode: the gradients (and the flag word) were written into the bucket by a replayed backward graph; sum over
        ranks.  The division by the world size and the optimiser steps are replayed from their own graph afterwards."""
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)

    def gate_sum(self, counts4: torch.Tensor) -> torch.Tensor:
        """Global gate (args.sgs_dp_global_gate): the ranks' [#correct learned, #train, #correct random, #train] summed in place, so
        that learned-vs-random is decided once for the union of the ranks' batches and every rank runs the SAME backward branch
        (with per-rank gates a step waits for the one rank whose gate picked the ten times longer learned backward).  Ranks whose
        partition is not sampled pass zeros."""
        if is_parallel():
            dist.all_reduce(counts4, op=dist.ReduceOp.SUM)
        return counts4

    def any_learned(self, learned_local: torch.Tensor) -> torch.Tensor:
        """Device-side: sum over ranks of this rank's 0/1 gate outcome (enqueue BEFORE the gate read-back)."""
        if is_parallel():
            dist.all_reduce(learned_local, op=dist.ReduceOp.SUM)
        return learned_local

    def sync(self, all_random: bool = False) -> None:
        """Average all .grad tensors across ranks (missing grads count as zeros); afterwards every
        parameter's .grad is a view of the shared bucket.  `all_random`: no rank took the learned branch this
        step, so the parameters without a gradient here have none on any rank -- they keep `.grad = None` and
        the optimisers skip them exactly as on one GPU (Adam with a zero gradient would still move them)."""
        if not is_parallel():
            return
        self._ensure(self.params[0].device)
        have_v, have_g, miss_v = [], [], []
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                miss_v.append(v)
            elif p.grad.data_ptr() != v.data_ptr():
                have_v.append(v)
                have_g.append(p.grad)
        if have_v:
            torch._foreach_copy_(have_v, have_g)
        if miss_v:
            torch._foreach_zero_(miss_v)
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        self.flat.div_(dist.get_world_size())
        missing = {id(p) for p in self.params if p.grad is None} if all_random else ()
        for p, v in zip(self.params, self.views):
            p.grad = None if id(p) in missing else v
