"""Mirror of the hot-path helpers of the reference's utils.py."""
from __future__ import annotations

import ctypes
import os
import random

import numpy as np
import torch

from . import ops


def calculate_f1(logits, labels, mask) -> float:
    """utils.py:163-169: sklearn micro-F1 of the argmax on masked rows == accuracy.  Computed on
    the device (sgs_masked_correct); only two ints cross to the host."""
    c = ops.masked_correct(logits, labels, mask).tolist()
    return c[0] / c[1] if c[1] else 0.0


def consistency_loss(edge_probs, edge_indices, node_embeddings):
    """utils.py:187-211: MSE(edge_probs, cos(emb[src], emb[dst]))."""
    N = node_embeddings.shape[0]
    dev = node_embeddings.device
    y = torch.zeros(N, dtype=torch.int64, device=dev)
    tm = torch.zeros(N, dtype=torch.bool, device=dev)
    total, _ = ops.edge_regularizers(edge_probs, node_embeddings, edge_indices, y, tm, 0.0, 1.0)
    return total


def fix_seeds(seed=42):
    """utils.py:82-89, plus the counter-based noise / dropout streams of this package."""
    from .sampling import manual_seed
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    manual_seed(seed)


# ------------------------------------------------------------------ segment profiler (utils.py:13-80) + rocTX ranges
class _RocTx:
    """roctxRangePushA / roctxRangePop of the rocprofv3 marker library (librocprofiler-sdk-roctx.so; no-ops unless a profiler is
    attached: `rocprofv3 --marker-trace --kernel-trace -- python3 ...` shows the ranges beside the kernels)."""
    _lib = False            # False: not looked for yet; None: not available

    @classmethod
    def lib(cls):
        if cls._lib is False:
            cls._lib = None
            root = os.environ.get("ROCM_PATH", "/opt/rocm")
            for name in ("librocprofiler-sdk-roctx.so", "libroctx64.so"):
                try:
                    L = ctypes.CDLL(os.path.join(root, "lib", name))
                    L.roctxRangePushA.argtypes = [ctypes.c_char_p]
                    L.roctxRangePushA.restype = ctypes.c_int
                    L.roctxRangePop.restype = ctypes.c_int
                    cls._lib = L
                    break
                except (OSError, AttributeError):
                    continue
        return cls._lib

    @classmethod
    def push(cls, name: str) -> None:
        L = cls.lib()
        if L is not None:
            L.roctxRangePushA(name.encode())

    @classmethod
    def pop(cls) -> None:
        L = cls.lib()
        if L is not None:
            L.roctxRangePop()


SEGMENTS = ("edge_mlp_pre", "edge_score", "gnn_forward", "backward")      # model.py:17-43,103-131,156-163; training_hybrid.py:22-27


class GpuMemoryProfiler:
    """Drop-in for the reference's `GpuMemoryProfiler` (utils.py:13-80; attached as `model.gpu_profiler` and
    `model.edge_prob_mlp.gpu_profiler`, main.py:116-119): `begin(name)` / `end(name)` around the four segments `SEGMENTS`, per-epoch
    summaries with the reference's keys.  Two things on top: every segment is also a rocTX range of the same name (visible in a
    rocprofv3 trace next to the kernels it launched), and `memory=False` keeps the ranges but skips the two device synchronisations
    per segment that the memory snapshots cost (the reference's epoch times under --gpu_profile are pessimistic for that reason,
    SURVEY.md section 6).  In HIP-graph mode a step's segments are replayed, not launched: the ranges then bracket the replays
    (`replay:G1`, `backward`)."""

    def __init__(self, enabled=False, device=None, memory=True, roctx=True):
        self.device = torch.device(device) if device is not None else torch.device("cuda")
        self.enabled = bool(enabled) and torch.cuda.is_available() and self.device.type == "cuda"
        self.memory, self.roctx = bool(memory), bool(roctx)
        self._epoch = None
        self._rows = {}            # epoch -> segment -> [(peak increase, allocated after, allocated increase)]
        self._open = {}

    def start_epoch(self, epoch):
        if self.enabled:
            self._epoch = epoch
            self._rows.setdefault(epoch, {})

    def end_epoch(self):
        if self.enabled:
            self._epoch = None
            self._open.clear()

    def _snapshot(self):
        torch.cuda.synchronize(self.device)
        return torch.cuda.max_memory_allocated(self.device), torch.cuda.memory_allocated(self.device)

    def begin(self, name):
        if not self.enabled or self._epoch is None:
            return
        if self.roctx:
            _RocTx.push(name)
        self._open[name] = self._snapshot() if self.memory else (0, 0)

    def end(self, name):
        if not self.enabled or self._epoch is None:
            return 0, 0
        was = self._open.pop(name, None)
        if was is None:
            return 0, 0
        peak, alloc = self._snapshot() if self.memory else (0, 0)
        if self.roctx:
            _RocTx.pop()
        row = (max(0, peak - was[0]), alloc, alloc - was[1])
        self._rows.setdefault(self._epoch, {}).setdefault(name, []).append(row)
        return row[0], row[1]

    def summarize_epoch(self, epoch):
        out = {}
        if not self.enabled:
            return out
        mb = float(1024 ** 2)
        for name, rows in self._rows.get(epoch, {}).items():
            if not rows:
                continue
            cols = list(zip(*rows))
            d = {"calls": len(rows)}
            for key, vals in (("peak_inc", cols[0]), ("alloc_after", cols[1]), ("alloc_inc", cols[2])):
                d[f"max_{key}_bytes"] = max(vals)
                d[f"max_{key}_mb"] = max(vals) / mb
                d[f"mean_{key}_mb"] = (sum(vals) / len(vals)) / mb
            out[name] = d
        return out


class segment:
    """`with segment(module_or_profiler, "edge_score"):` -- begin / end on the object's `gpu_profiler` when it has one (no-op otherwise)."""
    __slots__ = ("prof", "name")

    def __init__(self, owner, name):
        self.prof = owner if isinstance(owner, GpuMemoryProfiler) else getattr(owner, "gpu_profiler", None)
        self.name = name

    def __enter__(self):
        if self.prof is not None:
            self.prof.begin(self.name)
        return self

    def __exit__(self, *exc):
        if self.prof is not None:
            self.prof.end(self.name)
        return False
