"""Adam as one launch per parameter group (csrc/optim.hip): a drop-in for the two `torch.optim.Adam` objects the
reference builds (main.py:100, 122) with the same update rule, the same `state_dict` layout (`step`, `exp_avg`,
`exp_avg_sq` per parameter) and `capturable=True`, so `train(..., args.sgs_hipgraph=True)` records its steps inside
the backward graphs.

    optimizer_gnn = sgs_gnn_amd.FusedAdam([p for n, p in model.named_parameters() if 'gcn' in n], lr=args.lr)

Parameters whose `.grad` is None are skipped, exactly as torch does (that matters: the two optimisers overlap on
`edge_prob_mlp.gcn*`, and the random-wins branch leaves the scorer without gradients)."""
from __future__ import annotations

import ctypes

import torch

from . import _lib, ops


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, maximize=False):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("FusedAdam: invalid hyper-parameters")
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, maximize=maximize, capturable=True, amsgrad=False,
                        foreach=None, fused=None, differentiable=False)
        super().__init__(params, defaults)
        self._tickets = {}
        self._ticket_pool = None          # one zero-initialised int32 block per optimiser; a launch's ticket is a view of it

    def _init_state(self, p):
        if self._ticket_pool is None:
            # created with the state, i.e. BEFORE any capture: a ticket born inside a capture would be re-zeroed by a fill kernel on
            # every replay (5 us per optimiser step at partition scale)
            self._ticket_pool = torch.zeros(64, dtype=torch.int32, device=p.device)
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    @torch.no_grad()
    def step(self, closure=None, gate=None, gated=None):
        """`gate`: optional device float tensor [1]; tensors in `gated` (a set of parameters; None = every tensor of this
        optimiser) are left untouched while it reads 0 -- a device-side `if` with no host round trip."""
        loss = None
        gate_ptr = 0
        if gate is not None:
            if not gate.is_cuda or gate.dtype != torch.float32 or gate.numel() != 1:
                raise RuntimeError("FusedAdam: gate must be one float32 device word")
            gate_ptr = gate.data_ptr()
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.lib()
        kmax = L.sgs_adam_max_tensors()
        stream = ops._stream()
        for gi, group in enumerate(self.param_groups):
            plist = [p for p in group["params"] if p.grad is not None]
            if not plist:
                continue
            b1, b2 = group["betas"]
            for lo in range(0, len(plist), kmax):
                part = plist[lo:lo + kmax]
                words = []
                for p in part:
                    g = p.grad
                    if (p.dtype != torch.float32 or g.dtype != torch.float32 or not p.is_cuda or g.is_sparse or not p.is_contiguous()
                            or not g.is_contiguous()):
                        raise RuntimeError("FusedAdam: parameters and gradients must be dense, contiguous float32 HIP tensors")
                    st = self._init_state(p)
                    words += [p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(),
                              st["step"].data_ptr(), gate_ptr if (gated is None or p in gated) else 0]
                key = (gi, lo)
                tk = self._tickets.get(key)
                if tk is None:
                    if self._ticket_pool is not None and len(self._tickets) < self._ticket_pool.numel():
                        i = len(self._tickets)
                        tk = self._tickets[key] = self._ticket_pool[i:i + 1]
                    else:
                        tk = self._tickets[key] = torch.zeros(1, dtype=torch.int32, device=part[0].device)
                arr = (ctypes.c_int64 * len(words))(*words)
                _lib.check(L.sgs_adam_step(arr, len(part), float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                           float(group["weight_decay"]), int(bool(group["maximize"])), tk.data_ptr(), stream),
                           "sgs_adam_step")
                # the kernel wrote the parameters through raw pointers: tell autograd / version-keyed caches, as an in-place
                # torch op would have.  Not while a HIP graph is being captured: nothing executes then, and the capture goes on to
                # record the OTHER backward branch through the same saved tensors (only one of the two is replayed per step).
                if not torch.cuda.is_current_stream_capturing():
                    torch.autograd.graph.increment_version(part)
        return loss
