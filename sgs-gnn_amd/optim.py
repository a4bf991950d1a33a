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

    @staticmethod
    @torch.no_grad()
    def step_many(optimizers, tick=None):
        """`for o in optimizers: o.step()` as ONE launch (sgs_adam_step_multi): a tensor that several of the optimisers hold -- the
        reference's two Adam objects overlap on edge_prob_mlp.gcn*, main.py:100-109, 122 -- is updated once per optimiser, in order,
        while its elements are in registers.  `tick` = (loss_sum, loss, epoch): the launch also closes a replayed step (ops.loss_tick).
        Falls back to separate steps for anything the fused launch does not cover (a tensor in more than two optimisers)."""
        opts = list(optimizers)
        L = _lib.lib()
        order, states = [], {}
        for o in opts:
            if not isinstance(o, FusedAdam):
                raise RuntimeError("FusedAdam.step_many: every optimiser must be a FusedAdam")
            for group in o.param_groups:
                for p in group["params"]:
                    if p.grad is None:
                        continue
                    if id(p) not in states:
                        states[id(p)] = []
                        order.append(p)
                    states[id(p)].append((o, group))
        if any(len(v) > 2 for v in states.values()):
            for o in opts:
                o.step()
            if tick is not None:
                ops.loss_tick(*tick)
            return
        tick_ptrs = (None, None, None)
        if tick is not None:
            loss_sum, loss, epoch = tick
            tick_ptrs = (loss_sum.data_ptr(), loss.detach().reshape(1).data_ptr(), epoch.data_ptr())
        kmax = L.sgs_adam_multi_max_tensors()
        stream = ops._stream()
        first = opts[0]
        if not order:
            if tick is not None:
                ops.loss_tick(*tick)
            return
        for lo in range(0, len(order), kmax):
            part = order[lo:lo + kmax]
            words, hyper = [], []
            for p in part:
                g = p.grad
                if (p.dtype != torch.float32 or g.dtype != torch.float32 or not p.is_cuda or g.is_sparse or not p.is_contiguous()
                        or not g.is_contiguous()):
                    raise RuntimeError("FusedAdam: parameters and gradients must be dense, contiguous float32 HIP tensors")
                w = [p.data_ptr(), g.data_ptr(), p.numel()]
                h = []
                for o, group in states[id(p)]:
                    st = o._init_state(p)
                    w += [st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), st["step"].data_ptr(), 0]
                    b1, b2 = group["betas"]
                    h += [float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]), float(bool(group["maximize"]))]
                if len(states[id(p)]) == 1:
                    w += [0, 0, 0, 0]
                    h += [0.0] * 6
                words += w
                hyper += h
            key = ("multi", lo, tuple(id(o) for o in opts))
            tk = first._tickets.get(key)
            if tk is None:
                if first._ticket_pool is not None and len(first._tickets) < first._ticket_pool.numel():
                    i = len(first._tickets)
                    tk = first._tickets[key] = first._ticket_pool[i:i + 1]
                else:
                    tk = first._tickets[key] = torch.zeros(1, dtype=torch.int32, device=part[0].device)
            last = lo + kmax >= len(order)
            arr = (ctypes.c_int64 * len(words))(*words)
            hy = (ctypes.c_float * len(hyper))(*hyper)
            tp = tick_ptrs if last else (None, None, None)
            _lib.check(L.sgs_adam_step_multi(arr, hy, len(part), tk.data_ptr(), tp[0], tp[1], tp[2], stream), "sgs_adam_step_multi")
            if not torch.cuda.is_current_stream_capturing():
                torch.autograd.graph.increment_version(part)

