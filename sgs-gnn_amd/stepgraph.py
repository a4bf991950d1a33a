"""Opt-in HIP-graph replay of the per-partition training step (`args.sgs_hipgraph = True`).

Why: at partition scale (n ~ 1e3 nodes, E <= 5e5 edges) one hybrid step is ~70-150 kernel launches of a few microseconds each,
and the Python + launch time per step exceeds the GPU-busy time.  The C ABI never allocates or
synchronises, so a step's device work is capturable as is.

ONE capture serves every partition.  A captured graph freezes pointers and sizes, and partitions differ in both; instead of
one capture per partition (round 1: eager first visit, capture on the second, ~0.2 GiB of private pool each) the step is
recorded over a SLOT of static input buffers sized for the largest partition of the loader:

    slot     = x [Npad, F], y, train_mask, edge_index [2, Ecap], prob [Ecap], the partition's cached CSR (both orientations),
               and one device word `dims` = the live edge count
    staging  = ONE launch (sgs_stage_segments) copies a partition's resident arrays into the slot, pads the tails (zero rows for
               the nodes past N: isolated, unlabelled, outside every mask -- they change no sum; row pointers past N = E) and
               writes `dims`
    dynamic E: the two kernels of the step whose grids run over the candidate edges (sgs_edge_score_fwd, sgs_sample_topq) read
               the live E from `dims` (sgs_dyn_edges_set) and treat the captured E as capacity; everything after the draw is
               sized by q and Npad, which a run fixes; the CSR of a drawn subgraph is squeezed out of the staged parent CSR row
               by row.
    result   = every training step is a replay, from the first step of the first epoch on; graph memory is 2 slots x 2 kinds.

Two kinds of step, each captured once per slot:

    E_b >  q :  G0 = prior draw -> CSR of the random graph -> its unit normalisation      (parameter-independent prefix)
                G1 = scores -> learned draw -> CSR build -> learned / random encoders -> the two correct-counts -> publish
                     them to pinned host memory                                           (training.sampled_forward)
                the host polls the gate words (the step's one read-back)
                G2L = CE + reg1 + reg2, backward of the learned branch   |   G2R = CE, backward of the random branch
    E_b <= q :  G  = encoder on all edges, CE, backward
    every step's last launch adds its loss to the running sum and bumps the RNG epoch (sgs_loss_tick)

Two slots per kind (ping-pong) make the hand-over free: while step i runs out of slot A, the trainer names batch i+1
(`forward(batch, next_batch)`) and its staging copy -- and, for a sampled partition, its G0 -- run on a second stream into slot
B.  G0 takes its scratch from its own arena and its RNG epoch from its own device word, which the host writes before each
replay (a replayed step ticks the epoch exactly once, so the host knows the next value): the noise stream is identical to
replaying G0 in line (`SGS_SG_PREFETCH=0`).

Optimisers: `capturable` ones (sgs_gnn_amd.FusedAdam, or torch's with capturable=True) are recorded at the end of the
backward graphs -- their state is created before any capture; others are stepped eagerly after the replay, with `.grad` of
every parameter pointed at that graph's static gradient buffer (or None where the branch gives no gradient).
Data parallel (N > 1, FusedAdam): the backward graphs accumulate straight into GradSync's flat bucket (zeroed by the graph)
and add one "this rank learned" word; the trainer issues ONE all-reduce and replays g3, a single shared graph that divides
by the world size and steps both optimisers with the all-reduced word read on the device (FusedAdam's per-tensor gate).
Every rank replays from its first step on, so all ranks issue the same collectives in the same order whatever they hold.

Randomness: seeds are launch arguments and therefore frozen at capture; every replayed step ends by incrementing the
registered RNG epoch word (ops.set_rng_epoch_buffer) which all RNG-consuming kernels fold into their seed, so each replay
draws fresh Exp(1) noise and dropout masks.  The random stream therefore differs from eager mode's (same distributions);
parity tests run eager mode.  The per-step loss is accumulated on the device (`loss_sum`) and read once per epoch.
"""
from __future__ import annotations

import ctypes
import gc
import os
import time

import torch

from . import _lib, ops
from .data import Batch

_DEBUG = os.environ.get("SGS_SG_DEBUG", "")     # "fork": capture the random encoder on a second stream (measured slower)
_PREFETCH = os.environ.get("SGS_SG_PREFETCH", "1") != "0"
_E_ALIGN = 2048                                  # slot edge capacity granule (the sampler's chunk)


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


def _capture_context(model, params, device):
    """(capture stream, the parameters' AccumulateGrad nodes created on it) -- once per model; see StepGraphs.__init__."""
    ctx = getattr(model, "_sgs_capture_ctx", None)
    ids = tuple(id(p) for p in params)
    if ctx is None or ctx[2] != ids:
        gc.collect()                                   # retire autograd graphs (and with them older AccumulateGrad nodes) that only cycles keep alive
        stream = ctx[0] if ctx is not None else torch.cuda.Stream(device=device)
        with torch.cuda.stream(stream):
            acc = [p.view_as(p).grad_fn.next_functions[0][0] for p in params if p.requires_grad]
        ctx = model._sgs_capture_ctx = (stream, acc, ids)
    return ctx[0], ctx[1]


def _batch_key(batch):
    return (batch.x.data_ptr(), batch.edge_index.data_ptr(), batch.y.data_ptr(), batch.train_mask.data_ptr(),
            batch.prob.data_ptr() if getattr(batch, "prob", None) is not None else 0, batch.edge_index.shape[1],
            batch.x.shape[0])


def _round_up(v: int, m: int) -> int:
    return ((int(v) + m - 1) // m) * m


class _Slot:
    """Static input buffers of one captured step + its graphs."""
    __slots__ = ("sampled", "index", "npad", "ecap", "batch", "dims", "graph", "norm", "mask4", "canon", "mate", "g0", "g1", "g2l", "g2r", "cbuf", "loss",
                 "loss_l", "loss_r", "grads", "grads_l", "grads_r", "keep", "pre_event", "pre_epoch", "staged", "stage_event", "live")


class StepGraphs:
    """Per-model registry of the captured step (two slots per kind of step)."""

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
        # RNG epoch: 0 during warm-up, then the 1-based index of the replayed step (set to 1 by the first capture, ticked by each
        # replayed step's last launch)
        self.epoch_word = torch.zeros(1, dtype=torch.int64, device=self.device)
        self._epoch_started = False
        # Warm-up and captures share one side stream.  Autograd stamps every node with the stream that was current when the node
        # was created, and the engine synchronises a gradient's producer stream with its consumer's: a parameter whose
        # AccumulateGrad node was born on another stream (an eager backward on the default stream, say) would pull that stream
        # into the capture -- on ROCm 7.2 that ends in a segmentation fault inside hipStreamEndCapture.  So the capture stream is
        # created once per model, every parameter's AccumulateGrad node is created ON it, and the nodes are kept alive for the
        # model's lifetime: whatever runs a backward later, on whichever stream, reuses them.
        self.stream, self._grad_acc = _capture_context(model, self.params, self.device)
        # Second capture stream (SGS_SG_DEBUG=fork): the random encoder as a parallel branch beside the scorer.  Correct (tests pass
        # with it), but on ROCm 7.2 a two-branch graph costs 133 us of host time per launch instead of 24 us and the GPU time of the
        # segment does not drop (810 vs 790 us): the step got 5 % slower, so the capture stays single-stream.
        self.side = torch.cuda.Stream(device=self.device)
        # prefetch (module docstring): the stream staging copies and G0 replays run on when they are issued ahead, the epoch word
        # G0's kernels read, the host's mirror of `epoch_word` (value at the start of the next replayed step), and an event
        self.pre_stream = torch.cuda.Stream(device=self.device)
        self.epoch_pre = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.host_epoch = 0
        self.ev_main = torch.cuda.Event()
        self.last_pre = None                            # event of the most recent work issued on pre_stream
        self.one = torch.ones((), dtype=torch.float32, device=self.device)      # root gradient: saves autograd's ones_like fill per backward
        self.loss_sum = torch.zeros((), dtype=torch.float32, device=self.device)  # replayed steps add their loss here (read once per epoch)
        # gate read-back of a replayed step without a copy-engine round trip: G1 ends by publishing the counts to pinned,
        # device-mapped host memory (sgs_publish_to_host) and the trainer polls the sequence word
        self.host_gate = torch.zeros(8, dtype=torch.int32).pin_memory()
        self.host_gate_np = self.host_gate.numpy()
        self.host_gate_np[4] = -1                       # sequence word: differs from any epoch value a replay will publish
        self.slots = {True: [], False: []}              # sampled? -> [_Slot, _Slot]
        self.turn = {True: 0, False: 0}
        self.npad, self.ecap = 0, {True: 0, False: 0}   # capacities (nodes; candidate edges per kind)
        fc1 = getattr(getattr(model, "edge_prob_mlp", None), "fc1", None)
        self.pairs_ok = fc1 is not None and bool(_lib.lib().sgs_edge_score_paired_supported(int(fc1.weight.shape[0])))
        self.nfeat = None
        self.stage_cache = {}                           # (batch key, slot id) -> (descriptor array, n segments, dims array, keep-alive)
        # the fused scorer backward (ops._edge_score_backward_fused) needs edge lists sorted by source; a capture bakes the choice in, so
        # the slots' static edge_index carries the flag and a partition that is not sorted drops the captures made under it (forward())
        self.src_sorted = True
        # sparse node features (ops.FeatCSR; CitationFull-Cora's bag-of-words rows): when EVERY partition of the loader has one, the slots carry
        # static CSR buffers of this capacity, filled by the staging copy, and the captured first-layer products run over the non-zeros
        self.fcsr_cap = 0
        self.capture_seconds = 0.0
        self.captures = 0
        self.debug_keep = False                         # tests: keep static views of a replay's draws / outputs per slot
        self.seed_state = {}                            # kind -> (noise clock, dropout clock) its captures start from
        self.cfg = self._config_key()

    def _config_key(self):
        a = self.args
        return (self.pipeline, bool(a.conditional), bool(a.sparse_edge_mlp), a.reg1 == True, a.reg2 == True,   # noqa: E712
                float(a.regularizer1_coef), float(a.consist_reg_coef), float(a.degree_bias_coef), int(self.q),
                bool(self.use_checkpoint), tuple(p.data_ptr() for p in self.params), _opt_signature(self.optimizers))

    @classmethod
    def attach(cls, model, pipeline, args, criterion, q, use_checkpoint, optimizers=None, sync=None, loader=None):
        """`optimizers` = (optimizer_edge_prob, optimizer_gnn): when both are built with `capturable=True` their steps are recorded
        at the end of the backward graphs, so a replayed step needs no eager launch at all; otherwise the trainer steps them
        eagerly after each replay.  `loader`: when it can be scanned without being consumed (a list, a ResidentPartitions) the
        slots are sized for its largest partition up front; otherwise they grow on demand (a re-capture)."""
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
        if loader is not None:
            sg.reserve(loader)
        return sg

    def release(self):
        ops.set_rng_epoch_buffer(None)
        ops.drop_memos(self.model)          # a memo made during a capture points into that graph's pool

    # ------------------------------------------------------------------ capacities
    def reserve(self, loader) -> None:
        """Size the slots for the largest partition of `loader` (only if it can be scanned without consuming it)."""
        batches = getattr(loader, "batches", None)
        if batches is None and isinstance(loader, (list, tuple)):
            batches = loader
        if batches is None:
            return
        n, e = self.npad, dict(self.ecap)
        fcap, fall = 0, True
        for b in batches:
            if b is None or not hasattr(b, "edge_index"):
                continue
            E = int(b.edge_index.shape[1])
            n = max(n, int(b.x.shape[0]))
            e[E > self.q] = max(e[E > self.q], E)
            if b.x.is_cuda and E > self.q and self.src_sorted and not ops.src_sorted(b.edge_index):
                self.src_sorted = False                 # settled before the first capture (see forward())
            if b.x.is_cuda and E > 0:
                src = self._sources(b, want_norm=E <= self.q, want_pairs=E > self.q and self.pairs_ok)   # CSR (+ unit norm / mates) of every resident partition, once
                fc = src["fcsr"]
                fall = fall and fc is not None
                fcap = max(fcap, fc.nnz if fc is not None else 0)
        if fall and fcap > 0 and not any(self.slots.values()):
            self.fcsr_cap = max(self.fcsr_cap, _round_up(fcap, 1024))
        self._set_capacity(n, e)

    def _set_capacity(self, n, e) -> None:
        e = {k: (_round_up(v, _E_ALIGN) if v > 0 else 0) for k, v in e.items()}
        if n > self.npad or any(e[k] > self.ecap[k] for k in e):
            if any(self.slots.values()):
                torch.cuda.synchronize()
            self.npad = max(n, self.npad)
            self.ecap = {k: max(e[k], self.ecap[k]) for k in e}
            self.slots = {True: [], False: []}          # captures made for the smaller capacity are dropped
            self.stage_cache.clear()

    def _fits(self, batch) -> bool:
        E = int(batch.edge_index.shape[1])
        return int(batch.x.shape[0]) <= self.npad and 0 < E <= self.ecap[E > self.q] and (self.nfeat in (None, int(batch.x.shape[1])))

    def _fcsr_ok(self, batch) -> bool:
        """Does the partition fit the slots' static feature-CSR buffers (trivially true when the slots carry none)?"""
        if not self.fcsr_cap:
            return True
        fc = ops.feature_csr(batch.x, build=True)
        return fc is not None and fc.nnz <= self.fcsr_cap

    # ------------------------------------------------------------------ slots and staging
    def _new_slot(self, sampled: bool, index: int, like) -> _Slot:
        dev, N, Ecap = self.device, self.npad, self.ecap[sampled]
        F = int(like.x.shape[1])
        self.nfeat = F
        s = _Slot()
        for name in _Slot.__slots__:
            setattr(s, name, None)
        s.sampled, s.index, s.npad, s.ecap = sampled, index, N, Ecap
        z = dict(device=dev)
        x = torch.zeros(N, F, dtype=torch.float32, **z)
        fcs = None
        if self.fcsr_cap:
            fcs = ops.FeatCSR()
            fcs.N, fcs.F, fcs.nnz = N, F, self.fcsr_cap
            i32 = dict(dtype=torch.int32, device=dev)
            fcs.ptr, fcs.tptr = torch.zeros(N + 1, **i32), torch.zeros(F + 1, **i32)
            fcs.col, fcs.trow = torch.zeros(self.fcsr_cap, **i32), torch.zeros(self.fcsr_cap, **i32)
            fcs.val, fcs.tval = torch.zeros(self.fcsr_cap, dtype=torch.float32, **z), torch.zeros(self.fcsr_cap, dtype=torch.float32, **z)
        x._sgs_fcsr = (fcs, x._version)                 # (None: the slot's x is never scanned for sparsity -- its content changes per partition)
        y = torch.zeros(N, dtype=torch.int64, **z)
        s.mask4 = torch.zeros(_round_up(N, 4), dtype=torch.uint8, **z)
        tm = s.mask4[:N].view(torch.bool)
        ei = torch.zeros(2, Ecap, dtype=torch.int64, **z)
        prob = torch.zeros(Ecap, dtype=torch.float32, **z) if sampled else None
        s.batch = Batch(x=x, edge_index=ei, y=y, train_mask=tm, prob=prob)
        s.dims = torch.zeros(2, dtype=torch.int64, **z)             # live E; live number of canonical edges (paired scorer forward)
        # the slot's "cached parent graph": same object layout as ops.Graph, arrays filled by the staging copy
        g = ops.Graph.__new__(ops.Graph)
        g.edge_index, g.n_edges, g.N = ei, Ecap, N
        i32 = dict(dtype=torch.int32, device=dev)
        g.in_ptr, g.out_ptr = torch.zeros(N + 1, **i32), torch.zeros(N + 1, **i32)
        g.in_src, g.in_eid, g.out_dst, g.out_eid = (torch.zeros(max(Ecap, 1), **i32) for _ in range(4))
        g.loop_eid = torch.full((max(N, 1),), -1, **i32)
        ei._sgs_graph, ei._sgs_graph_version = g, ei._version
        ei._sgs_src_sorted = (self.src_sorted, ei._version)
        s.graph = g
        if sampled and self.pairs_ok:
            # mates of the paired scorer forward: static buffers filled by the staging copy, their live length in dims[1]
            s.canon, s.mate = torch.zeros(max(Ecap, 1), **i32), torch.full((max(Ecap, 1),), -1, **i32)
            ei._sgs_pairs = (s.canon, s.mate, ei._version)
        if not sampled:
            nm = ops.Norm()
            nm.graph, nm.w, nm.handle = g, None, None
            f32 = dict(dtype=torch.float32, device=dev)
            nm.dis, nm.loopw = torch.ones(N, **f32), torch.ones(N, **f32)
            nm.what_in, nm.what_out = torch.zeros(max(Ecap, 1), **f32), torch.zeros(max(Ecap, 1), **f32)
            nm.what_loop = torch.ones(N, **f32)
            g._norm_unit = nm
            s.norm = nm
        s.pre_event, s.stage_event = torch.cuda.Event(), torch.cuda.Event()
        return s

    @staticmethod
    def _sources(batch, want_norm=False, want_pairs=False):
        """The partition's resident arrays the staging copy reads (built once per partition, cached on the batch): its tensors,
        the CSR of its edge list and -- for the unsampled step -- the unit normalisation; the train mask as whole 4-byte words.
        Launches kernels that take scratch from the MAIN arena: call it on the main stream only (never on the prefetch stream,
        where it would scribble over the scratch of the step in flight)."""
        src = getattr(batch, "_sgs_stage_src", None)
        if src is not None and want_norm and src["norm"] is None:
            src["norm"] = ops.gcn_norm(src["graph"], None)
        if src is not None and want_pairs and src["pairs"] is None:
            src["pairs"] = ops.get_pairs(batch.edge_index, int(batch.x.shape[0]), build=True)
        if src is None:
            N = int(batch.x.shape[0])
            g = ops.get_graph(batch.edge_index, N)
            m4 = torch.zeros(_round_up(N, 4), dtype=torch.uint8, device=batch.x.device)
            m4[:N] = ops._u8(batch.train_mask)
            src = dict(graph=g, mask4=m4, x=batch.x.contiguous(), y=batch.y.contiguous(), norm=ops.gcn_norm(g, None) if want_norm else None,
                       pairs=ops.get_pairs(batch.edge_index, N, build=True) if want_pairs else None, fcsr=ops.feature_csr(batch.x, build=True))
            try:
                batch._sgs_stage_src = src
            except Exception:
                pass
        return src

    def _stage_desc(self, batch, slot: _Slot):
        key = (_batch_key(batch), id(slot))
        hit = self.stage_cache.get(key)
        if hit is not None:
            return hit
        src = self._sources(batch, want_norm=not slot.sampled, want_pairs=slot.sampled and slot.canon is not None)
        g, sg = src["graph"], slot.graph
        N, E = int(batch.x.shape[0]), int(batch.edge_index.shape[1])
        Np, Ec = slot.npad, slot.ecap
        F = int(batch.x.shape[1])
        sb = slot.batch
        segs = [(src["x"], sb.x, N * F * 4, Np * F * 4, 0),
                (src["y"], sb.y, N * 8, Np * 8, 0),
                (src["mask4"], slot.mask4, src["mask4"].numel(), slot.mask4.numel(), 0),
                (g.in_ptr, sg.in_ptr, (N + 1) * 4, (Np + 1) * 4, E),
                (g.out_ptr, sg.out_ptr, (N + 1) * 4, (Np + 1) * 4, E),
                (g.in_src, sg.in_src, E * 4, E * 4, 0),
                (g.out_dst, sg.out_dst, E * 4, E * 4, 0),
                (g.in_eid, sg.in_eid, E * 4, E * 4, 0),
                (g.out_eid, sg.out_eid, E * 4, E * 4, 0),
                (g.loop_eid, sg.loop_eid, N * 4, Np * 4, 0xFFFFFFFF)]
        if slot.sampled:
            ei = batch.edge_index
            if not ei.is_contiguous():
                raise RuntimeError("sgs_gnn_amd: batch.edge_index must be contiguous")
            segs += [(ei.data_ptr(), sb.edge_index.data_ptr(), E * 8, E * 8, 0),
                     (ei.data_ptr() + E * 8, sb.edge_index.data_ptr() + Ec * 8, E * 8, E * 8, 0),
                     (batch.prob.contiguous(), sb.prob, E * 4, E * 4, 0)]
            M = 0
            if slot.canon is not None:
                canon, mate = src["pairs"]
                M = int(canon.numel())
                segs += [(canon, slot.canon, M * 4, M * 4, 0), (mate, slot.mate, E * 4, E * 4, 0)]
        else:
            nm, sn = src["norm"], slot.norm
            segs += [(nm.what_in, sn.what_in, E * 4, E * 4, 0),
                     (nm.what_out, sn.what_out, E * 4, E * 4, 0),
                     (nm.what_loop, sn.what_loop, N * 4, Np * 4, 0x3F800000)]        # an isolated padded node: deg 1, loop weight 1.0f
        if self.fcsr_cap:
            fc, fs = src["fcsr"], sb.x._sgs_fcsr[0]
            if fc is None or fc.nnz > self.fcsr_cap:
                raise RuntimeError("sgs_gnn_amd: partition does not fit the slots' sparse-feature buffers")     # (forward() re-captures before this)
            nz = fc.nnz
            segs += [(fc.ptr, fs.ptr, (N + 1) * 4, (Np + 1) * 4, nz), (fc.col, fs.col, nz * 4, nz * 4, 0), (fc.val, fs.val, nz * 4, nz * 4, 0),
                     (fc.tptr, fs.tptr, (F + 1) * 4, (F + 1) * 4, 0), (fc.trow, fs.trow, nz * 4, nz * 4, 0), (fc.tval, fs.tval, nz * 4, nz * 4, 0)]
        words = []
        keep = []
        for s_, d_, nb_s, nb_d, pad in segs:
            sp = s_ if isinstance(s_, int) else s_.data_ptr()
            dp = d_ if isinstance(d_, int) else d_.data_ptr()
            keep.append((s_, d_))
            words += [sp, dp, nb_s, nb_d, pad]
        arr = (ctypes.c_int64 * len(words))(*words)
        dims = (ctypes.c_int64 * 2)(E, M if slot.sampled else 0)
        if len(self.stage_cache) >= 8192:
            # a loader that produces fresh batch objects every epoch must not pin them all: the entries keep their sources alive
            for k in list(self.stage_cache)[:4096]:
                del self.stage_cache[k]
        hit = self.stage_cache[key] = (arr, len(segs), dims, keep)
        return hit

    def _stage(self, batch, slot: _Slot) -> None:
        """The partition's arrays -> the slot's static buffers, on the current stream (one launch)."""
        arr, n, dims, _ = self._stage_desc(batch, slot)
        L = _lib.lib()
        _lib.check(L.sgs_stage_segments(arr, n, slot.dims.data_ptr(), dims, 2, ops._stream()), "sgs_stage_segments")
        slot.live, slot.staged, slot.pre_epoch = batch, None, None      # (`live` also keeps the partition's tensors alive: its key stays unique)

    def _pick(self, sampled: bool, like, avoid=None) -> _Slot:
        sl = self.slots[sampled]
        while len(sl) < 2:
            sl.append(self._new_slot(sampled, len(sl), like))
        i = self.turn[sampled]
        if avoid is not None and sl[i] is avoid:
            i ^= 1
        self.turn[sampled] = i ^ 1
        return sl[i]

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

    def null_step(self):
        """Data-parallel step without a batch (training._null_step): zero gradients and flag word into the bucket, the step's one
        all-reduce, then the shared optimiser graph -- the same collectives and the same update as the ranks that had a batch."""
        if self.g3 is None:
            self._capture_g3()
        self.sync._ensure(self.device)
        self.sync.flat.zero_()
        self.sync.all_reduce_bucket()
        self.g3.replay()

    def _capture(self, slot: _Slot) -> None:
        # The cyclic collector must not run inside a capture: it may finalise objects whose destructors call into
        # the runtime (an old CUDAGraph, a stream, an event), which is illegal while a stream is capturing.  Collect
        # first (this also retires autograd nodes left over from eager steps), then hold the collector off.
        was_enabled = gc.isenabled()
        gc.collect()
        gc.disable()
        t0 = time.perf_counter()
        cur = torch.cuda.current_stream()
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            ops.workspace_handover(self.device)          # the capture stream is ordered after the main stream: it owns the arenas now
        ops.set_dyn_edges(slot.dims)        # kernels over the candidate edges read the live count from the slot (min rule, sgs_hip.h)
        from .model import _DropoutClock
        from .sampling import _NoiseClock
        clocks = None
        try:
            self._warm_up(slot)
            # Seeds are launch arguments, frozen at capture.  Both slots of a kind must freeze the SAME seeds, or a step's noise
            # would depend on which slot happened to host it (in-line vs prefetched hand-over pick slots in a different order).
            now = (_NoiseClock.tick, _DropoutClock.tick)
            first = self.seed_state.setdefault(slot.sampled, now)
            if first != now:
                clocks = now
                _NoiseClock.tick, _DropoutClock.tick = first
            self._capture_segments(slot)
        finally:
            if clocks is not None:
                _NoiseClock.tick, _DropoutClock.tick = clocks
            ops.set_dyn_edges(None)
            ops.drop_memos(self.model)      # memos made inside the capture hold graph-pool memory a later replay overwrites
            self._clear_grads()
            cur.wait_stream(self.stream)
            torch.cuda.synchronize()
            ops.workspace_handover(self.device)          # ... and back (everything on the capture stream has finished)
            self.capture_seconds += time.perf_counter() - t0
            self.captures += 1
            if was_enabled:
                gc.enable()

    def _warm_up(self, slot: _Slot) -> None:
        """One eager pass over the slot on the capture stream before recording it: lazy code-object loads and first-use library
        calls are not capturable, and BOTH backward branches must have run once.  No optimiser step: the parameters are untouched."""
        from .training import _ce, learned_loss, sampled_forward
        b, a = slot.batch, self.args
        ops.drop_memos(self.model)
        ops.new_memo_scope()
        with torch.cuda.stream(self.stream):
            if not slot.sampled:
                _ce(self.criterion, self.model(b, b.edge_index), b).backward()
            else:
                st = sampled_forward(self.pipeline, a, self.model, b, self.q, self.use_checkpoint)
                if st.random_out is not None:
                    _ce(self.criterion, st.random_out, b).backward(retain_graph=True)
                    self._clear_grads()
                learned_loss(a, self.criterion, st, b).backward()
        self._clear_grads()
        torch.cuda.synchronize()

    def _capture_segments(self, c: _Slot) -> None:
        from .training import _ce, learned_loss, sampled_forward, sampled_prefix
        if not self._epoch_started:
            self.epoch_word.fill_(1)
            self.host_epoch = 1
            self._epoch_started = True
        a, batch = self.args, c.batch
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
                self._steps_and_tick((self.optimizers[1],) if (self.optimizers is not None and not self.dp) else (), c.loss)   # optimizer_gnn (training_hybrid.py:161)
            c.grads = self._grads()
            c.loss = c.loss.detach()
            self._clear_grads()
            return
        pre, pool = None, None
        if (a.conditional or a.sparse_edge_mlp) and _DEBUG != "fork":
            # G0: the parameter-independent head, with its own scratch arena and epoch word so that it may run beside the main stream
            c.g0 = torch.cuda.CUDAGraph()
            ops.set_rng_epoch_buffer(self.epoch_pre)
            try:
                with ops.workspace_slot(1), torch.cuda.graph(c.g0, stream=self.stream, capture_error_mode=self.capture_mode):
                    pre = sampled_prefix(a, batch, self.q)
            finally:
                ops.set_rng_epoch_buffer(self.epoch_word)
            pool = c.g0.pool()
        with torch.cuda.graph(c.g1, stream=self.stream, pool=pool, capture_error_mode=self.capture_mode):
            st = sampled_forward(self.pipeline, a, self.model, batch, self.q, self.use_checkpoint,
                                 side_stream=self.side if _DEBUG == "fork" else None,   # measured: a forked capture is SLOWER here
                                 prefix=pre, gate_publish=(self.epoch_word, self.host_gate))
        pool = c.g1.pool()
        c.cbuf = st.cbuf
        if self.debug_keep:
            # static views of the replay's own draws and outputs (graph-pool memory: valid until this slot is replayed again);
            # tests recompute the step eagerly from these
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
            # optimizer_edge_prob, then optimizer_gnn (:136-137), and the step's closing tick: one launch with FusedAdam
            self._steps_and_tick(tuple(self.optimizers) if (self.optimizers is not None and not self.dp) else (), loss_l)
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
                self._steps_and_tick((self.optimizers[1],) if (self.optimizers is not None and not self.dp) else (), loss_r)   # optimizer_gnn only (:141)
            c.grads_r = self._grads()
            c.loss_r = loss_r.detach()
            self._clear_grads()

    def _steps_and_tick(self, opts, loss) -> None:
        """The optimiser steps a backward graph ends with, then `loss_sum += loss; epoch += 1` (the RNG epoch: the next replay draws
        fresh noise).  FusedAdam optimisers: ONE launch for all of it (sgs_adam_step_multi); others: their own step() + sgs_loss_tick."""
        from .optim import FusedAdam
        if opts and all(isinstance(o, FusedAdam) for o in opts):
            FusedAdam.step_many(opts, tick=(self.loss_sum, loss, self.epoch_word))
            return
        for o in opts:
            o.step()
        ops.loss_tick(self.loss_sum, loss, self.epoch_word)

    # ------------------------------------------------------------------ one step
    def _set_grads(self, grads):
        for i, p in enumerate(self.params):
            p.grad = grads.get(i)

    def replay_g1(self, c: _Slot):
        """G0 (if the step has one) and G1 in line on the current stream, G0 with the device's current epoch.  For tools and tests
        that replay segments of a staged slot by hand."""
        if c.g0 is not None:
            main = torch.cuda.current_stream()
            if self.last_pre is not None:
                main.wait_event(self.last_pre)
            self.epoch_pre.copy_(self.epoch_word)
            c.g0.replay()
            c.pre_epoch = None
        c.g1.replay()

    def _prefetch(self, nxt, after, avoid: _Slot) -> None:
        """Batch `nxt`'s staging copy -- and, for a sampled partition, its G0 for the step whose epoch will be host_epoch + 1 --
        on the prefetch stream, once the main-stream work up to `after` is done."""
        sampled = int(nxt.edge_index.shape[1]) > self.q
        slot = self._pick(sampled, nxt, avoid=avoid)
        if slot.g1 is None:
            return                                     # not captured yet: the next forward() stages and captures in line
        if (_batch_key(nxt), id(slot)) not in self.stage_cache:
            # first hand-over of this partition: its CSR may still have to be built, with scratch from the main arena -> on the main
            # stream, and the copy waits for it (once per partition; reserve() has done it for loaders it could scan)
            self._stage_desc(nxt, slot)
            after = torch.cuda.Event()
            after.record(torch.cuda.current_stream())
        self.pre_stream.wait_event(after)
        with torch.cuda.stream(self.pre_stream):
            self._stage(nxt, slot)
            slot.pre_epoch = None
            if slot.g0 is not None:
                self.epoch_pre.fill_(self.host_epoch + 1)
                slot.g0.replay()
                slot.pre_epoch = self.host_epoch + 1
            slot.stage_event.record(self.pre_stream)
        slot.staged = _batch_key(nxt)
        self.last_pre = slot.stage_event

    def forward(self, batch, next_batch=None):
        """Runs the step up to the gate (E_b > q) or completely (E_b <= q) and returns the handle the trainer
        finishes the step with: `h.sampled`, `h.cbuf` (gate counts, device int32[5]) and `h.backward(learned)`.
        `next_batch` (optional): the batch of the following step; its staging copy and prefix are issued ahead."""
        if self.src_sorted and int(batch.edge_index.shape[1]) > self.q and not ops.src_sorted(batch.edge_index):
            # (cached on the partition's tensor: one read-back the first time a partition is seen)
            torch.cuda.synchronize()
            self.src_sorted = False
            self.slots = {True: [], False: []}          # captured for source-sorted partitions: recorded again without the fused backward
            self.stage_cache.clear()
        if not self._fcsr_ok(batch):
            torch.cuda.synchronize()
            self.fcsr_cap = 0                           # a partition with dense (or more) features: the slots go back to the library GEMM
            self.slots = {True: [], False: []}
            self.stage_cache.clear()
        if not self._fits(batch):
            E = int(batch.edge_index.shape[1])
            if E == 0 or (self.nfeat not in (None, int(batch.x.shape[1]))):
                return _EagerHandle(self, batch)       # degenerate / foreign batch: launched eagerly
            e = dict(self.ecap)
            e[E > self.q] = max(e[E > self.q], int(E * 1.25) if any(self.slots.values()) else E)
            self._set_capacity(max(self.npad, int(batch.x.shape[0])), e)
        sampled = int(batch.edge_index.shape[1]) > self.q
        key = _batch_key(batch)
        main = torch.cuda.current_stream()
        c = next((s for s in self.slots[sampled] if s.staged == key and s.g1 is not None), None)
        prefix_ahead = False
        if c is not None:                              # staged (and prefixed) ahead during the previous step
            main.wait_event(c.stage_event)
            prefix_ahead = c.g0 is not None and c.pre_epoch is not None and c.pre_epoch == self.host_epoch
        else:
            # a partition that is still sitting in a slot (cluster_loader = [data], main.py:67: the same batch every step) is not copied again
            c = next((s for s in self.slots[sampled] if s.live is not None and s.g1 is not None and _batch_key(s.live) == key), None)
            if c is None:
                c = self._pick(sampled, batch)
                if self.last_pre is not None:
                    main.wait_event(self.last_pre)     # in line: after whatever the prefetch stream was last given
                self._stage(batch, c)
                if c.g1 is None:
                    self._capture(c)
        c.staged = None
        if c.g0 is not None and not prefix_ahead:
            if self.last_pre is not None:
                main.wait_event(self.last_pre)
            self.epoch_pre.copy_(self.epoch_word)
            c.g0.replay()
        c.pre_epoch = None
        want_next = _PREFETCH and next_batch is not None and self._fits(next_batch)
        if want_next:
            self.ev_main.record(main)                  # the next partition's hand-over may start once everything before THIS G1 is done
        seq0 = int(self.host_gate_np[4])
        prof = getattr(self.model, "gpu_profiler", None)
        if prof is not None:
            prof.begin("replay:G1" if c.sampled else "replay:G")      # the reference's forward segments are inside this replay
        c.g1.replay()
        if prof is not None:
            prof.end("replay:G1" if c.sampled else "replay:G")
        if want_next:
            self._prefetch(next_batch, self.ev_main, avoid=c)
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

    def memory_bytes(self) -> int:
        """Static slot buffers held by this registry (inputs; the graphs' private pools come on top)."""
        n = 0
        for sl in self.slots.values():
            for s in sl:
                ts = [s.batch.x, s.batch.y, s.mask4, s.batch.edge_index, s.batch.prob, s.graph.in_ptr, s.graph.out_ptr, s.graph.in_src,
                      s.graph.in_eid, s.graph.out_dst, s.graph.out_eid, s.graph.loop_eid]
                if s.norm is not None:
                    ts += [s.norm.what_in, s.norm.what_out, s.norm.what_loop, s.norm.dis, s.norm.loopw]
                n += sum(t.numel() * t.element_size() for t in ts if t is not None)
        return n


class _ReplayHandle:
    __slots__ = ("sg", "c", "sampled", "cbuf", "opt_in_graph", "seq0", "loss_on_device", "dp_in_handle")
    replayed = True                                    # training.py: no zero_grad() needed, backward() attaches the captured gradient buffers

    def __init__(self, sg, c, seq0):
        self.sg, self.c, self.sampled, self.cbuf = sg, c, c.sampled, c.cbuf
        self.opt_in_graph = sg.optimizers is not None      # the backward graphs (or, data-parallel, g3) hold the optimiser steps
        self.seq0 = seq0
        self.loss_on_device = True                         # the graphs add their loss to sg.loss_sum
        self.dp_in_handle = sg.dp                          # data parallel: backward() issues the bucket all-reduce and replays g3

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
        elif learned or c.g2r is None:
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
    """A batch the slots cannot take (no edges, another feature width): the same segments launched eagerly.
    Data-parallel graph mode: this rank's peers replay their steps, i.e. they issue [gate sum,] ONE all-reduce of the flat bucket
    (gradients + "this rank learned" word) and then the shared optimiser graph g3.  The eager step must join exactly those
    collectives -- a different sequence (a flag all-reduce, GradSync.sync) would pair up with the wrong calls on the other ranks --
    so its backward writes the gradients and the flag into the bucket, all-reduces it and replays g3 as well."""
    __slots__ = ("sg", "batch", "sampled", "cbuf", "st", "opt_in_graph", "loss_on_device", "dp_in_handle")
    replayed = False

    def __init__(self, sg, batch):
        from .training import sampled_forward
        self.sg, self.batch = sg, batch
        self.opt_in_graph = sg.dp                          # data parallel: g3 steps the optimisers
        self.dp_in_handle = sg.dp
        self.loss_on_device = False
        self.sampled = batch.edge_index.shape[1] > sg.q
        self.cbuf, self.st = None, None
        if self.sampled:
            self.st = sampled_forward(sg.pipeline, sg.args, sg.model, batch, sg.q, sg.use_checkpoint)
            self.cbuf = self.st.cbuf

    def gate_counts(self):
        return self.cbuf.tolist()[:4]

    def backward(self, learned):
        from .training import _ce, learned_loss
        sg, a, batch, st = self.sg, self.sg.args, self.batch, self.st
        if not self.sampled:
            loss = _ce(sg.criterion, sg.model(batch, batch.edge_index), batch)
        else:
            loss = learned_loss(a, sg.criterion, st, batch) if (learned or st.random_out is None) else _ce(sg.criterion, st.random_out, batch)
        loss.backward()
        self.st = None
        if sg.dp:
            sy = sg.sync
            sy._ensure(sg.device)
            if sg.g3 is None:
                sg._capture_g3()
            have_v, have_g, miss_v = [], [], []
            for p, v in zip(sy.params, sy.views):
                if p.grad is None:
                    miss_v.append(v)
                else:
                    have_v.append(v)
                    have_g.append(p.grad)
            if have_v:
                torch._foreach_copy_(have_v, have_g)
            if miss_v:
                torch._foreach_zero_(miss_v)
            sy.flag.fill_(1.0 if (self.sampled and (learned or st.random_out is None)) else 0.0)
            sg._clear_grads()                              # g3 reads the bucket's views, which it was captured over
            sy.all_reduce_bucket()
            sg.g3.replay()
        return loss.detach()
