"""Opt-in HIP-graph replay of the per-partition training step (`args.sgs_hipgraph = True`).

Why: at partition scale (n ~ 1e3 nodes, E <= 5e5 edges) one hybrid step is ~70-150 kernel launches of a few
microseconds each, and the Python + launch time per step exceeds the GPU-busy time.
The C ABI never allocates or synchronises, so a step's device work is capturable as is.  Per partition
(keyed on the batch's tensors) the step is recorded once into HIP graphs and replayed afterwards:

    E_b >  q :  G1 = prior draw -> scores -> learned draw -> CSR build -> learned / random encoders ->
                     the two correct-counts -> publish them to pinned host memory   (training.sampled_forward)
                the host polls the gate words (the step's one read-back; eager mode does a 20-byte copy instead)
                G2L = CE + reg1 + reg2, backward of the learned branch   |   G2R = CE, backward of the random branch
    E_b <= q :  G  = encoder on all edges, CE, backward
    every step's last launch adds its loss to the running sum and bumps the RNG epoch (sgs_loss_tick)

Prefix prefetch: the head of G1 -- prior draw, CSR of the random graph, its unit normalisation: 12 dependent launches,
~90 us at 350 k edges -- depends on the partition and the noise alone, not on the parameters.  It is captured as its own
graph G0 (scratch from its own arena, RNG epoch from its own device word) and, when the trainer names the next batch
(`forward(batch, next_batch)`), replayed for the NEXT partition on a second stream while the current partition's G1 / gate
read-back / backward are in flight; G1 of that partition then only waits for an event.  The epoch G0 folds into its noise is
written by the host (a replayed step ticks the epoch exactly once, so the host knows the next value), which keeps the
noise stream identical to the unsplit capture.  `SGS_SG_PREFETCH=0` keeps G0 in line on the main stream.

Optimisers: `capturable` ones (sgs_gnn_amd.FusedAdam, or torch's with capturable=True) are recorded at the end of the
backward graphs -- their state is created before any capture, a state tensor born inside a capture would be reset by
every replay; others are stepped eagerly after the replay, with `.grad` of every parameter pointed at that graph's static
gradient buffer (or None where the branch gives no gradient: Adam must skip those, as in eager mode).
Data parallel (N > 1, FusedAdam): the backward graphs accumulate straight into GradSync's flat bucket (zeroed by the graph)
and add one "this rank learned" word; the trainer issues ONE all-reduce and replays g3, a single shared graph that divides
by the world size and steps both optimisers with the all-reduced word read on the device (FusedAdam's per-tensor gate).

Randomness: seeds are launch arguments and therefore frozen at capture; every replayed step ends by
incrementing the registered RNG epoch word (ops.set_rng_epoch_buffer) which all RNG-consuming kernels fold
into their seed, so each replay draws fresh Exp(1) noise and dropout masks.  The random stream therefore
differs from eager mode's (same distributions); parity tests run eager mode.

First visit of a partition runs eagerly on the capture stream (warm-up: lazy code-object loads, CSR cache, workspace
growth, BOTH backward branches), the second visit captures (and replays), later visits only replay.  The per-step loss
is accumulated on the device (`loss_sum`) and read once per epoch.
"""
from __future__ import annotations

import gc
import os

import torch

from . import ops

_DEBUG = os.environ.get("SGS_SG_DEBUG", "")     # "fork": capture the random encoder on a second stream (measured slower)
_PREFETCH = os.environ.get("SGS_SG_PREFETCH", "1") != "0"


class _Captured:
    __slots__ = ("key", "sampled", "g0", "g1", "g2l", "g2r", "cbuf", "loss", "loss_l", "loss_r", "grads", "grads_l", "grads_r",
                 "keep", "pre_event", "pre_epoch")


def _capturable(opt) -> bool:
    return all(bool(g.get("capturable", False)) for g in opt.param_groups)


def _opt_signature(optimizers):
    """Hyper-parameters a captured optimiser step bakes in (python scalars become kernel arguments)."""
    if optimizers is None:
        return None
    sig = []
    for o in optimizers:
        for g in o.param_groups:
            sig.append((id(o), tuple(sorted((k, v) for k, v in g.items() if isinstance(v, (int, float, bool, tuple, type(None)))))))
    return tuple(sig)


def _ensure_optimizer_state(opt) -> bool:
    """Optimiser state must exist BEFORE a step is captured: state created inside a capture (zeros for the moments, the step
    counter) would be re-created -- i.e. reset -- by every replay.  Returns False for optimisers this module cannot prepare."""
    from .optim import FusedAdam
    if isinstance(opt, FusedAdam):
        for grp in opt.param_groups:
            for p in grp["params"]:
                opt._init_state(p)
        return True
    if type(opt) is torch.optim.Adam:
        for grp in opt.param_groups:
            for p in grp["params"]:
                st = opt.state[p]
                if len(st) == 0:
                    st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)      # capturable layout
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    if grp.get("amsgrad", False):
                        st["max_exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return True
    return False


def _batch_key(batch):
    return (batch.x.data_ptr(), batch.edge_index.data_ptr(), batch.y.data_ptr(), batch.train_mask.data_ptr(),
            batch.prob.data_ptr() if getattr(batch, "prob", None) is not None else 0, batch.edge_index.shape[1],
            batch.x.shape[0])


class StepGraphs:
    """Per-model registry of captured partition steps."""

    def __init__(self, model, pipeline, args, criterion, q, use_checkpoint, optimizers=None, sync=None):
        self.model = model
        self.optimizers = optimizers      # (optimizer_edge_prob, optimizer_gnn) when their steps are captured too
        # data-parallel graph mode (N > 1 ranks, FusedAdam optimisers): the backward graphs write the gradients straight into
        # GradSync's flat bucket, the trainer all-reduces it, and ONE more graph (g3) averages and steps both optimisers with
        # the "did any rank learn" word read on the device -- one collective and no host round trip per step besides the gate
        self.sync = sync
        self.g3 = None
        # With a process group alive, other threads (the collective watchdog) issue runtime queries at any time; "global" capture
        # mode would turn one of those into a capture error, so captures only police their own thread then.
        import torch.distributed as dist
        self.capture_mode = "thread_local" if (dist.is_available() and dist.is_initialized()) else "global"
        self.pipeline = pipeline
        self.args = args
        self.criterion = criterion
        self.q = q
        self.use_checkpoint = use_checkpoint
        self.params = [p for p in model.parameters()]
        self.device = self.params[0].device
        # RNG epoch: 0 during the eager warm-up visits, then the 1-based index of the replayed step (set to 1 by the first
        # capture, ticked by each replayed step's last launch)
        self.epoch_word = torch.zeros(1, dtype=torch.int64, device=self.device)
        self._epoch_started = False
        # warm-up visits and captures share one side stream: autograd stamps every node (AccumulateGrad included)
        # with the stream it was created on, and a capture must not meet nodes from the legacy default stream
        self.stream = torch.cuda.Stream(device=self.device)
        # Second capture stream (SGS_SG_DEBUG=fork): the random encoder as a parallel branch beside the scorer.  Correct (tests pass
        # with it), but on ROCm 7.2 a two-branch graph costs 133 us of host time per launch instead of 24 us and the GPU time of the
        # segment does not drop (810 vs 790 us): the step got 5 % slower, so the capture stays single-stream.
        self.side = torch.cuda.Stream(device=self.device)
        # prefix prefetch (module docstring): stream G0 replays run on when they are issued ahead, the epoch word G0's kernels
        # read, the host's mirror of `epoch_word` (value at the start of the next replayed step), and two events
        self.pre_stream = torch.cuda.Stream(device=self.device)
        self.epoch_pre = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.host_epoch = 0
        self.ev_main = torch.cuda.Event()
        self.last_pre = None                            # event of the most recent G0 issued on pre_stream
        self.one = torch.ones((), dtype=torch.float32, device=self.device)      # root gradient: saves autograd's ones_like fill per backward
        self.loss_sum = torch.zeros((), dtype=torch.float32, device=self.device)  # replayed steps add their loss here (read once per epoch)
        # gate read-back of a replayed step without a copy-engine round trip: G1 ends by publishing the counts to pinned,
        # device-mapped host memory (sgs_publish_to_host) and the trainer polls the sequence word
        self.host_gate = torch.zeros(8, dtype=torch.int32).pin_memory()
        self.host_gate_np = self.host_gate.numpy()
        self.host_gate_np[4] = -1                       # sequence word: differs from any epoch value a replay will publish
        self.table = {}          # batch key -> _Captured | "seen"
        self.cfg = self._config_key()

    def _config_key(self):
        a = self.args
        return (self.pipeline, bool(a.conditional), bool(a.sparse_edge_mlp), a.reg1 == True, a.reg2 == True,   # noqa: E712
                float(a.regularizer1_coef), float(a.consist_reg_coef), float(a.degree_bias_coef), int(self.q),
                bool(self.use_checkpoint), tuple(p.data_ptr() for p in self.params), _opt_signature(self.optimizers))

    @classmethod
    def attach(cls, model, pipeline, args, criterion, q, use_checkpoint, optimizers=None, sync=None):
        """`optimizers` = (optimizer_edge_prob, optimizer_gnn): when both are built with `capturable=True` (and the
        run is single-process) their steps are recorded at the end of the backward graphs, so a replayed step
        needs no eager launch at all; otherwise the trainer steps them eagerly after each replay."""
        if optimizers is not None and not (all(_capturable(o) for o in optimizers) and all(_ensure_optimizer_state(o) for o in optimizers)):
            optimizers = None
        if sync is not None:
            from .optim import FusedAdam
            if optimizers is None or not all(isinstance(o, FusedAdam) for o in optimizers):
                optimizers, sync = None, None              # N > 1 without FusedAdam: eager all-reduce + eager optimiser steps
        sg = getattr(model, "_sgs_stepgraphs", None)
        if sg is not None and (sg.sync is None) != (sync is None):
            sg = None
        fresh = cls(model, pipeline, args, criterion, q, use_checkpoint, optimizers, sync) if sg is None else None
        if sg is not None:
            sg.args, sg.criterion = args, criterion
            old_opts = sg.optimizers
            sg.optimizers = optimizers
            if ((sg.pipeline, sg.q, sg.use_checkpoint) != (pipeline, q, use_checkpoint) or sg._config_key() != sg.cfg
                    or (old_opts is None) != (optimizers is None)):
                fresh = cls(model, pipeline, args, criterion, q, use_checkpoint, optimizers, sync)   # settings changed: drop old captures
        if fresh is not None:
            sg = model._sgs_stepgraphs = fresh
        ops.set_rng_epoch_buffer(sg.epoch_word)
        ops.pin_workspaces(True)
        return sg

    def release(self):
        ops.set_rng_epoch_buffer(None)
        ops.drop_memos(self.model)          # a memo made during a capture points into that graph's pool

    def null_step(self):
        """Data-parallel step without a batch (training._null_step): zero gradients and flag word into the bucket, the step's one
        all-reduce, then the shared optimiser graph -- the same collectives and the same update as the ranks that had a batch."""
        if self.g3 is None:
            self._capture_g3()
        self.sync._ensure(self.device)
        self.sync.flat.zero_()
        self.sync.all_reduce_bucket()
        self.g3.replay()

    # ------------------------------------------------------------------ capture
    def _grads(self):
        return {i: p.grad for i, p in enumerate(self.params) if p.grad is not None}

    def _clear_grads(self):
        for p in self.params:
            p.grad = None

    @property
    def dp(self) -> bool:
        return self.sync is not None

    def _bind_bucket(self):
        """DP graph mode: every parameter's .grad is its view of the flat bucket, so a captured backward accumulates in place."""
        self.sync._ensure(self.device)
        for p, v in zip(self.sync.params, self.sync.views):
            p.grad = v

    def _capture_g3(self):
        """flat /= world; optimizer_edge_prob.step() unless no rank learned; optimizer_gnn.step() with the scorer's tensors
        skipped unless some rank learned -- all decided on the device from the all-reduced flag word."""
        import torch.distributed as dist
        sy, (oe, og) = self.sync, self.optimizers
        self._bind_bucket()
        scorer_params = {p for grp in oe.param_groups for p in grp["params"]}
        self.g3 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g3, stream=self.stream, capture_error_mode=self.capture_mode):
            sy.flat.div_(float(dist.get_world_size()))
            oe.step(gate=sy.flag)
            og.step(gate=sy.flag, gated=scorer_params)
        self._clear_grads()

    def _capture(self, batch, key) -> _Captured:
        # The cyclic collector must not run inside a capture: it may finalise objects whose destructors call into
        # the runtime (an old CUDAGraph, a stream, an event), which is illegal while a stream is capturing.  Collect
        # first (this also retires autograd nodes left over from eager steps), then hold the collector off.
        was_enabled = gc.isenabled()
        gc.collect()
        gc.disable()
        try:
            return self._capture_segments(batch, key)
        finally:
            ops.drop_memos(self.model)      # memos made inside the capture hold graph-pool memory a later replay overwrites
            if was_enabled:
                gc.enable()

    def _capture_segments(self, batch, key) -> _Captured:
        from .training import _ce, learned_loss, sampled_forward, sampled_prefix
        if not self._epoch_started:
            self.epoch_word.fill_(1)
            self.host_epoch = 1
            self._epoch_started = True
        a = self.args
        c = _Captured()
        for name in _Captured.__slots__:
            setattr(c, name, None)
        c.key = key
        c.sampled = batch.edge_index.shape[1] > self.q
        self._clear_grads()
        torch.cuda.synchronize()
        c.g1 = torch.cuda.CUDAGraph()
        ops.drop_memos(self.model)                     # no memoised x W^T from an eager step may leak into a capture
        ops.new_memo_scope()
        if self.dp and self.g3 is None:
            self._capture_g3()
        if not c.sampled:
            if self.dp:
                self._bind_bucket()
            with torch.cuda.graph(c.g1, stream=self.stream, capture_error_mode=self.capture_mode):
                if self.dp:
                    self.sync.flat.zero_()                         # gradients of this step + flag word (0: nobody learned here)
                out = self.model(batch, batch.edge_index)
                c.loss = _ce(self.criterion, out, batch)
                c.loss.backward(gradient=self.one)
                if self.optimizers is not None and not self.dp:
                    self.optimizers[1].step()                      # optimizer_gnn (training_hybrid.py:161)
                ops.loss_tick(self.loss_sum, c.loss, self.epoch_word)     # + RNG epoch: the next replay draws fresh noise
            c.grads = self._grads()
            c.loss = c.loss.detach()
            self._clear_grads()
            return c
        pre, pool = None, None
        if (a.conditional or a.sparse_edge_mlp) and _DEBUG != "fork":
            # G0: the parameter-independent head, with its own scratch arena and epoch word so that it may run beside the main stream
            c.g0 = torch.cuda.CUDAGraph()
            c.pre_event = torch.cuda.Event()
            ops.set_rng_epoch_buffer(self.epoch_pre)
            try:
                with ops.workspace_slot(1), torch.cuda.graph(c.g0, stream=self.stream, capture_error_mode=self.capture_mode):
                    pre = sampled_prefix(a, batch, self.q)
            finally:
                ops.set_rng_epoch_buffer(self.epoch_word)
            pool = c.g0.pool()
        with torch.cuda.graph(c.g1, stream=self.stream, pool=pool, capture_error_mode=self.capture_mode):
            st = sampled_forward(self.pipeline, a, self.model, batch, self.q, self.use_checkpoint,
                                 side_stream=self.side if _DEBUG == "fork" else None,   # measured: a forked capture is SLOWER here (below)
                                 prefix=pre)
            if st.cbuf is not None:
                ops.publish_to_host(st.cbuf, 4, self.epoch_word, self.host_gate)
        pool = c.g1.pool()
        c.cbuf = st.cbuf
        # static views of the replay's own draws and outputs (private-pool memory is never reused after the
        # capture, so holding them costs nothing); tests recompute the step eagerly from these
        c.keep = dict(rsei=st.rsei, eid=st.smp.eid, sampled_edge_index=st.sampled_edge_index,
                      edge_probs_full=st.edge_probs_full.detach(), w=st.edge_probs_for_loss.detach(),
                      learned_out=st.learned_out.detach(),
                      random_out=None if st.random_out is None else st.random_out.detach())
        c.g2l = torch.cuda.CUDAGraph()
        if self.dp:
            self._bind_bucket()
        with torch.cuda.graph(c.g2l, stream=self.stream, pool=pool, capture_error_mode=self.capture_mode):
            if self.dp:
                self.sync.flat.zero_()
            loss_l = learned_loss(a, self.criterion, st, batch)
            loss_l.backward(gradient=self.one, retain_graph=st.random_out is not None)
            if self.dp:
                self.sync.flag.fill_(1.0)                          # this rank's gate chose "learned"
            elif self.optimizers is not None:
                self.optimizers[0].step()                          # optimizer_edge_prob, then optimizer_gnn (:136-137)
                self.optimizers[1].step()
            ops.loss_tick(self.loss_sum, loss_l, self.epoch_word)
        c.grads_l = self._grads()
        c.loss_l = loss_l.detach()
        self._clear_grads()
        c.g2r = None
        if st.random_out is not None:
            c.g2r = torch.cuda.CUDAGraph()
            if self.dp:
                self._bind_bucket()
            with torch.cuda.graph(c.g2r, stream=self.stream, pool=pool, capture_error_mode=self.capture_mode):
                if self.dp:
                    self.sync.flat.zero_()
                loss_r = _ce(self.criterion, st.random_out, batch)
                loss_r.backward(gradient=self.one)
                if self.optimizers is not None and not self.dp:
                    self.optimizers[1].step()                      # optimizer_gnn only (:141)
                ops.loss_tick(self.loss_sum, loss_r, self.epoch_word)
            c.grads_r = self._grads()
            c.loss_r = loss_r.detach()
            self._clear_grads()
        return c

    # ------------------------------------------------------------------ one step
    def _set_grads(self, grads):
        for i, p in enumerate(self.params):
            p.grad = grads.get(i)

    def replay_g1(self, c):
        """G0 (if the step has one) and G1 in line on the current stream, G0 with the device's current epoch -- what a replay
        did before the prefix was split off.  For tools and tests that replay segments by hand."""
        if c.g0 is not None:
            main = torch.cuda.current_stream()
            if self.last_pre is not None:
                main.wait_event(self.last_pre)
            self.epoch_pre.copy_(self.epoch_word)
            c.g0.replay()
            c.pre_epoch = None
        c.g1.replay()

    def _issue_prefix(self, c, epoch: int, after) -> None:
        """G0 of `c` on the prefix stream, for the step whose epoch will be `epoch`, once the main-stream work up to `after` is done."""
        self.pre_stream.wait_event(after)
        with torch.cuda.stream(self.pre_stream):
            self.epoch_pre.fill_(epoch)
            c.g0.replay()
            c.pre_event.record(self.pre_stream)
        c.pre_epoch = epoch
        self.last_pre = c.pre_event

    def forward(self, batch, next_batch=None) -> "StepHandle":
        """Runs the step up to the gate (E_b > q) or completely (E_b <= q) and returns the handle the trainer
        finishes the step with: `h.sampled`, `h.cbuf` (gate counts, device int32[5]) and `h.backward(learned)`.
        `next_batch` (optional): the batch of the following step; its parameter-independent prefix is issued ahead."""
        key = _batch_key(batch)
        c = self.table.get(key)
        if c is None:                                  # first visit: eager, on the capture stream
            self.table[key] = "seen"
            return _EagerHandle(self, batch)
        if c == "seen":
            c = self.table[key] = self._capture(batch, key)
        main = torch.cuda.current_stream()
        if c.g0 is not None:
            if c.pre_epoch is not None and c.pre_epoch == self.host_epoch:
                main.wait_event(c.pre_event)           # issued ahead during the previous step
            else:                                      # in line: after whatever the prefix stream was last given
                if self.last_pre is not None:
                    main.wait_event(self.last_pre)
                self.epoch_pre.copy_(self.epoch_word)
                c.g0.replay()
            c.pre_epoch = None
        if _PREFETCH and next_batch is not None:
            self.ev_main.record(main)                  # the next partition's prefix may start once everything before THIS G1 is done
        seq0 = int(self.host_gate_np[4])
        c.g1.replay()
        if _PREFETCH and next_batch is not None:
            cn = self.table.get(_batch_key(next_batch))
            if isinstance(cn, _Captured) and cn is not c and cn.g0 is not None:
                self._issue_prefix(cn, self.host_epoch + 1, self.ev_main)
        return _ReplayHandle(self, c, seq0)

    def step(self, batch, epoch=0):
        """Single-process convenience: forward, gate read-back, backward.  Returns (loss, learned_won | None)."""
        h = self.forward(batch)
        if not h.sampled:
            loss = h.backward(None)
            if self.optimizers is not None and not h.opt_in_graph:
                self.optimizers[1].step()
            return loss, None
        won = True
        if h.cbuf is not None:
            cnt = h.gate_counts()                      # the step's one host read-back (gate)
            won = cnt[0] > cnt[2]
        loss = h.backward(won)
        if self.optimizers is not None and not h.opt_in_graph:
            if won:
                self.optimizers[0].step()
            self.optimizers[1].step()
        return loss, won


class _ReplayHandle:
    __slots__ = ("sg", "c", "sampled", "cbuf", "opt_in_graph", "seq0", "loss_on_device")

    def __init__(self, sg, c, seq0):
        self.sg, self.c, self.sampled, self.cbuf = sg, c, c.sampled, c.cbuf
        self.opt_in_graph = sg.optimizers is not None      # the backward graphs (or, data-parallel, g3) hold the optimiser steps
        self.seq0 = seq0
        self.loss_on_device = True                         # the graphs add their loss to sg.loss_sum

    def gate_counts(self):
        """[#correct learned, #train, #correct random, #train] of this replay: polls the pinned gate words G1 publishes
        (falls back to a stream synchronisation if the sequence word has not moved after a while)."""
        g = self.sg.host_gate_np
        for _ in range(200000):
            if g[4] != self.seq0:
                return [int(g[0]), int(g[1]), int(g[2]), int(g[3])]
        torch.cuda.current_stream().synchronize()
        if g[4] == self.seq0:
            raise RuntimeError("sgs_gnn_amd: the replayed step did not publish its gate counts")
        return [int(g[0]), int(g[1]), int(g[2]), int(g[3])]

    def backward(self, learned):
        c, sg = self.c, self.sg
        if not c.sampled:                              # the single graph already ran forward + backward
            loss, grads = c.loss, c.grads
        elif learned:
            c.g2l.replay()
            loss, grads = c.loss_l, c.grads_l
        else:
            c.g2r.replay()
            loss, grads = c.loss_r, c.grads_r
        sg.host_epoch += 1                             # every backward / unsampled graph ends with sgs_loss_tick
        if sg.dp:
            # gradients + flag word sit in the flat bucket: one all-reduce, then the shared graph that averages and steps
            sg.sync.all_reduce_bucket()
            sg.g3.replay()
        else:
            sg._set_grads(grads)
        return loss


class _EagerHandle:
    """First visit of a partition: the same segments launched eagerly on the capture stream (warm-up)."""
    __slots__ = ("sg", "batch", "sampled", "cbuf", "st", "cur", "opt_in_graph", "loss_on_device")

    def __init__(self, sg, batch):
        from .training import sampled_forward
        self.sg, self.batch = sg, batch
        self.opt_in_graph = False
        self.loss_on_device = False
        self.sampled = batch.edge_index.shape[1] > sg.q
        self.cbuf, self.st = None, None
        self.cur = torch.cuda.current_stream()
        sg.stream.wait_stream(self.cur)
        if self.sampled:
            with torch.cuda.stream(sg.stream):
                self.st = sampled_forward(sg.pipeline, sg.args, sg.model, batch, sg.q, sg.use_checkpoint)
            self.cur.wait_stream(sg.stream)
            self.cbuf = self.st.cbuf

    def gate_counts(self):
        return self.cbuf.tolist()[:4]

    def backward(self, learned):
        from .training import _ce, learned_loss
        sg, a, batch, st = self.sg, self.sg.args, self.batch, self.st
        sg.stream.wait_stream(self.cur)
        with torch.cuda.stream(sg.stream):
            if not self.sampled:
                loss = _ce(sg.criterion, sg.model(batch, batch.edge_index), batch)
            else:
                # warm-up must touch every kernel / library code path either capture will record (lazy code-object
                # loads and first-use attribute calls are not capturable): run the branch the gate rejected first
                if st.random_out is not None:
                    other = _ce(sg.criterion, st.random_out, batch) if learned else learned_loss(a, sg.criterion, st, batch)
                    other.backward(retain_graph=True)
                    sg._clear_grads()
                loss = learned_loss(a, sg.criterion, st, batch) if learned else _ce(sg.criterion, st.random_out, batch)
            loss.backward()
        self.cur.wait_stream(sg.stream)
        self.st = None
        return loss.detach()
