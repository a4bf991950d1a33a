"""Mirror of the reference's epoch trainers: training.py (dispatcher), training_hybrid.py,
training_straight_through.py and training_two_pass.py -- same `train(...)` signature and return
tuple, same control flow (gate, losses, which optimisers step), with the device work of every
step done by libsgs_hip.so.

Differences that do not change results (DESIGN.md "host loop"):
  * boolean-mask indexing (`edge_index[:, mask]`, `probs[mask]`, `out[train_mask]`) is replaced by
    the sampler's compacted outputs / masked kernels: no `nonzero` host syncs;
  * the F1 gate compares two on-device correct-counts (one 16-byte read-back per step, which the
    data-dependent choice of optimiser steps needs anyway); `loss.item()` is accumulated on the
    device and read once per epoch; reg1's `.item() > 1` test is a device-side predicate.

Test hooks (never set by main.py): `args._sgs_noise = {"prior": [E] , "sample": [E]}` feeds explicit
Exp(1) noise to the two draws; `args._sgs_trace = {}` receives the step's intermediates.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from .dist import GradSync, is_parallel
from .sampling import draw_learned, draw_prior, random_edge_sampling
from .utils import segment

_PIPELINES = ("two_pass", "straight_through", "hybrid")


def _fused_ce_ok(criterion) -> bool:
    return (type(criterion) is nn.CrossEntropyLoss and criterion.weight is None and criterion.reduction == "mean"
            and criterion.ignore_index == -100 and getattr(criterion, "label_smoothing", 0.0) == 0.0)


def _ce(criterion, out, batch):
    if _fused_ce_ok(criterion):
        return ops.masked_cross_entropy(out, batch.y, batch.train_mask)
    return criterion(out[batch.train_mask], batch.y[batch.train_mask])      # user-supplied criterion: run as given


def _backward(model, loss):
    """training_hybrid.py:22-27: loss.backward() inside the profiler's "backward" segment (a rocTX range, utils.GpuMemoryProfiler)."""
    with segment(model, "backward"):
        loss.backward()


def _has_train_nodes(batch) -> bool:
    flag = getattr(batch, "_sgs_has_train", None)
    if flag is None:
        flag = bool(batch.train_mask.any())
        try:
            batch._sgs_has_train = flag
        except Exception:
            pass
    return flag


class SampledForward:
    """Everything the sampled step's forward leaves behind for the gate and the two possible backwards."""
    __slots__ = ("rsei", "rs", "edge_probs_full", "smp", "sampled_edge_index", "edge_probs_for_loss", "learned_out",
                 "random_out", "cbuf")


def sampled_prefix(args, batch, q, noise=None):
    """The parameter-independent head of a sampled step (training_hybrid.py:44-48): the prior-only draw, the CSR of the random
    graph (squeezed out of the partition's cached CSR) and its unit normalisation.  Depends on the partition and the noise
    alone, so a captured step can run it for the NEXT partition while the current one is still in flight (stepgraph.py).
    Returns the draw, or None when the configuration has no prior draw."""
    if not (args.conditional or args.sparse_edge_mlp):
        return None
    noise = noise or {}
    rs = draw_prior(batch.prob, batch.edge_index, q, noise=noise.get("prior"))
    graph_r = ops.get_subgraph(batch.edge_index, batch.x.shape[0], rs)
    ops.gcn_norm(graph_r, None)                                       # cached on the graph: every unweighted layer over it reuses it
    return rs


def sampled_forward(pipeline, args, model, batch, q, use_checkpoint=False, noise=None, side_stream=None, prefix=None,
                    gate_publish=None) -> SampledForward:
    """training_hybrid.py:44-101 (ST 41-90, TP 41-92) up to the gate's inputs: prior draw, pass-1 scores, learned
    draw, encoder over the learned graph, encoder over the random graph, the two correct-counts (device side).
    No host read-back in here, so the whole segment can be captured into a HIP graph (stepgraph.py).
    `side_stream` (graph capture only): the encoder over the random graph depends on the prior draw alone, so it is issued
    on a second stream right after that draw and joins before the learned encoder -- in the captured graph it becomes a
    parallel branch beside the scorer instead of ~4 more dependent launches on the critical path.  (Kept as an option:
    measured slower on ROCm 7.2, see stepgraph.py.)  `gate_publish` = (sequence word, pinned host int32[5]): the launch that finishes
    the gate counts also writes them to the host."""
    noise = noise or {}
    st = SampledForward()
    N = batch.x.shape[0]
    scorer = model.edge_prob_mlp

    st.rsei, st.rs = None, None
    if args.conditional or args.sparse_edge_mlp:                      # K0: prior-only draw (`prefix`: already made by sampled_prefix)
        rs = prefix if prefix is not None else sampled_prefix(args, batch, q, noise)
        st.rsei, st.rs = rs.edge_index, rs
    st.random_out = None
    forked = side_stream is not None and args.conditional and st.rsei is not None
    if forked:
        ops.gcn_norm(ops.get_graph(st.rsei, N), None)                 # CSR + unit normalisation of the random graph: shared, built before the fork
        main_stream = torch.cuda.current_stream()
        side_stream.wait_stream(main_stream)
        with torch.cuda.stream(side_stream), ops.workspace_slot(2):      # beside the main stream: its own scratch arena
            ops.workspace_handover(batch.x.device, slot=2)
            st.random_out = model(batch, st.rsei)

    # pass 1: score every edge (hybrid / ST with grad, two-pass without)
    if pipeline == "two_pass":
        with torch.no_grad():
            st.edge_probs_full = scorer(batch.x, batch.edge_index, st.rsei).squeeze()
    elif pipeline == "hybrid":
        st.edge_probs_full = scorer(batch.x, batch.edge_index, st.rsei, use_checkpoint=use_checkpoint).squeeze()
    else:
        st.edge_probs_full = scorer(batch.x, batch.edge_index, st.rsei).squeeze()
    pass1_active = getattr(scorer, "last_active", None)

    # K2+K3: learned draw on detached probabilities, compacted columns in edge order
    st.smp = smp = draw_learned(batch.prob, st.edge_probs_full, batch.edge_index, q, args.degree_bias_coef,
                                noise=noise.get("sample"), want_p=(pipeline == "hybrid"))
    st.sampled_edge_index = smp.edge_index
    graph_s = ops.get_subgraph(batch.edge_index, N, smp)

    if pipeline == "hybrid":
        # edge_probs_full[mask]: gradient reaches only the q sampled entries
        if pass1_active is not None:
            pass1_active.set(smp.eid, graph_s)
        st.edge_probs_for_loss = ops.select_sampled(st.edge_probs_full, smp.eid, smp.p, pass1_active)    # = edge_probs_full[mask], gathered by the draw
    elif pipeline == "straight_through":
        st.edge_probs_for_loss = ops.st_weights(st.edge_probs_full, batch.prob, args.degree_bias_coef, smp.stats, smp.eid)
    else:
        # pass 3: re-score the sampled edges with grad; encoder over the learned graph
        st.edge_probs_for_loss = scorer(batch.x, smp.edge_index).squeeze()
    if forked:
        main_stream.wait_stream(side_stream)                          # join (also publishes the memoised x W^T of the GNN's first layer)
    st.learned_out = model(batch, smp.edge_index, st.edge_probs_for_loss)

    st.cbuf = None
    if args.conditional:
        if st.random_out is None:
            st.random_out = model(batch, st.rsei)
        # both counts (and, under graph capture, their hand-over to the polling host: gate_publish) in two launches, no zero fill
        st.cbuf = ops.gate_counts(st.learned_out, st.random_out, batch.y, batch.train_mask, publish=gate_publish)
    return st


def learned_loss(args, criterion, st: SampledForward, batch):
    """training_hybrid.py:105-133: CE + reg1 (BCE on same-class labels) + reg2 (cosine consistency)."""
    c1 = args.regularizer1_coef if args.reg1 == True else 0.0      # noqa: E712 (as the reference)
    c2 = args.consist_reg_coef if args.reg2 == True else 0.0       # noqa: E712
    if (c1 != 0.0 or c2 != 0.0) and _fused_ce_ok(criterion):
        # the three terms as one autograd node (ten launches -> six)
        return ops.hybrid_loss(st.learned_out, batch.y, batch.train_mask, st.edge_probs_for_loss, st.sampled_edge_index, c1, c2)[0]
    loss = _ce(criterion, st.learned_out, batch)
    if c1 != 0.0 or c2 != 0.0:
        reg, _ = ops.edge_regularizers(st.edge_probs_for_loss, st.learned_out, st.sampled_edge_index, batch.y,
                                       batch.train_mask, c1, c2)
        loss = loss + reg
    return loss


def train(args, epoch, max_epoch, model, optimizer_gnn, optimizer_edge_prob, optimizer, criterion, cluster_loader,
          q=500, alternate_frequency=1):
    """training.py:6-49: dispatch on args.pipeline (default two_pass)."""
    pipeline = getattr(args, "pipeline", "two_pass")
    if pipeline not in _PIPELINES:
        pipeline = "two_pass"
    return _train(pipeline, args, epoch, max_epoch, model, optimizer_gnn, optimizer_edge_prob, optimizer, criterion,
                  cluster_loader, q)


def prepare_step_graphs(args, model, optimizer_gnn, optimizer_edge_prob, criterion, cluster_loader, q=500):
    """Optional, `args.sgs_hipgraph` only: record the step's HIP graphs before the first epoch instead of inside it (train() does
    it on demand otherwise).  The slots are sized for the largest partition of `cluster_loader`; a representative batch of each
    kind (E_b > q, E_b <= q) is staged and the step is captured over both slots of that kind.  Nothing is trained: no optimiser
    step runs (a capture records, it does not execute; the warm-up pass discards its gradients).  Returns the seconds spent."""
    from .stepgraph import StepGraphs
    pipeline = getattr(args, "pipeline", "two_pass")
    if pipeline not in _PIPELINES:
        pipeline = "two_pass"
    sync = None
    if is_parallel():
        sync = getattr(model, "_sgs_gradsync", None)
        if sync is None:
            sync = model._sgs_gradsync = GradSync(model.parameters())
    model.train()
    sg = StepGraphs.attach(model, pipeline, args, criterion, q, bool(getattr(args, "hybrid_checkpoint", False)),
                           optimizers=(optimizer_edge_prob, optimizer_gnn), sync=sync, loader=cluster_loader)
    try:
        before = sg.capture_seconds
        batches = getattr(cluster_loader, "batches", cluster_loader)
        for kind in (True, False):
            rep = next((b for b in batches if (b.edge_index.shape[1] > q) == kind and _has_train_nodes(b)), None)
            if rep is None:
                continue
            rep = rep.to(args.device)
            for _ in range(2):
                slot = sg._pick(kind, rep)
                if slot.g1 is None:
                    sg._stage(rep, slot)
                    sg._capture(slot)
        return sg.capture_seconds - before
    finally:
        sg.release()


def train_hybrid(args, epoch, max_epoch, model, optimizer_gnn, optimizer_edge_prob, optimizer, criterion, cluster_loader,
                 q=500, alternate_frequency=1):
    """training_hybrid.train (training_hybrid.py:7-8): same positional / keyword signature (`alternate_frequency` is dead there too)."""
    return _train("hybrid", args, epoch, max_epoch, model, optimizer_gnn, optimizer_edge_prob, optimizer, criterion, cluster_loader, q)


def train_straight_through(args, epoch, max_epoch, model, optimizer_gnn, optimizer_edge_prob, optimizer, criterion, cluster_loader,
                           q=500, alternate_frequency=1):
    """training_straight_through.train (training_straight_through.py:7-8)."""
    return _train("straight_through", args, epoch, max_epoch, model, optimizer_gnn, optimizer_edge_prob, optimizer, criterion,
                  cluster_loader, q)


def train_two_pass(args, epoch, max_epoch, model, optimizer_gnn, optimizer_edge_prob, optimizer, criterion, cluster_loader,
                   q=500, alternate_frequency=1):
    """training_two_pass.train (training_two_pass.py:7-8)."""
    return _train("two_pass", args, epoch, max_epoch, model, optimizer_gnn, optimizer_edge_prob, optimizer, criterion, cluster_loader, q)


def _train(pipeline, args, epoch, max_epoch, model, optimizer_gnn, optimizer_edge_prob, optimizer, criterion,
           cluster_loader, q):
    device = args.device
    mode = args.mode
    use_checkpoint = bool(getattr(args, "hybrid_checkpoint", False))
    if pipeline == "hybrid" and epoch == 0:
        print(f"[hybrid] checkpoint={'on' if use_checkpoint else 'off'}")       # training_hybrid.py:12-13
    model.train()
    noise = getattr(args, "_sgs_noise", None) or {}
    trace = getattr(args, "_sgs_trace", None)
    sync = None
    if is_parallel():                       # N > 1: average gradients over ranks once per step (dist.py)
        sync = getattr(model, "_sgs_gradsync", None)
        if sync is None:
            sync = model._sgs_gradsync = GradSync(model.parameters())
    graphs = None
    if getattr(args, "sgs_hipgraph", False) and mode == 'learned' and not noise and trace is None and _fused_ce_ok(criterion):
        from .stepgraph import StepGraphs               # opt-in: replay captured HIP graphs of each partition's step
        graphs = StepGraphs.attach(model, pipeline, args, criterion, q, use_checkpoint,
                                   optimizers=(optimizer_edge_prob, optimizer_gnn), sync=sync, loader=cluster_loader)

    try:
        return _epoch_loop(pipeline, args, epoch, max_epoch, model, optimizer_gnn, optimizer_edge_prob, optimizer,
                           criterion, cluster_loader, q, device, mode, use_checkpoint, noise, trace, sync, graphs)
    finally:
        if graphs is not None:
            graphs.release()


def _with_lookahead(loader, on: bool):
    """(batch, following batch or None) pairs; without `on`, (batch, None) and the loader is consumed exactly as before."""
    if not on:
        for b in loader:
            yield b, None
        return
    it = iter(loader)
    try:
        cur = next(it)
    except StopIteration:
        return
    for nxt in it:
        yield cur, nxt
        cur = nxt
    yield cur, None


def _null_step(args, mode, model, optimizer_gnn, optimizer_edge_prob, optimizer, sync, graphs, device):
    """A data-parallel step on a rank that has no batch left: joins exactly the collectives a real step issues, contributes zero
    gradients and applies the same averaged update as the other ranks, so replicas stay identical."""
    if mode != 'learned':
        for p in sync.params:
            p.grad = None
        sync.sync()
        optimizer.step()
        return
    if graphs is not None and graphs.dp:                         # replayed steps: [gate sum] -> bucket all-reduce -> shared optimiser graph
        if args.conditional and bool(getattr(args, "sgs_dp_global_gate", False)):
            sync.gate_sum(torch.zeros(4, dtype=torch.int32, device=device))
        graphs.null_step()
        return
    optimizer_edge_prob.zero_grad()
    optimizer_gnn.zero_grad()
    flag = sync.any_learned(torch.zeros(1, dtype=torch.int32, device=device))
    some_learned = int(flag.item()) > 0
    if not some_learned:
        # nobody learned: exactly the GNN-only tensors carry gradients on the other ranks (random branch / unsampled step);
        # zeros here, so that this rank's Adam moves them with the same averaged gradient
        scorer = {id(p) for g in optimizer_edge_prob.param_groups for p in g["params"]}
        for g in optimizer_gnn.param_groups:
            for p in g["params"]:
                if id(p) not in scorer:
                    p.grad = torch.zeros_like(p)
    sync.sync(all_random=not some_learned)
    if some_learned:
        optimizer_edge_prob.step()
    optimizer_gnn.step()


_ALWAYS_ZERO_GRAD = False       # A/B switch (tools/host_ab.py): zero both optimisers at the top of every step, replayed or not


def _epoch_loop(pipeline, args, epoch, max_epoch, model, optimizer_gnn, optimizer_edge_prob, optimizer, criterion,
                cluster_loader, q, device, mode, use_checkpoint, noise, trace, sync, graphs):
    total_loss = None
    temperature = 1.0
    condtional_update = 0
    total_update = 0
    h = None
    if graphs is not None:
        graphs.loss_sum.zero_()
    loader, n_null = cluster_loader, 0
    if sync is not None:
        # N > 1: every step ends in collectives, so all ranks must run the same number of steps.  Shards may differ in length
        # (P % world != 0) and a rank may skip a batch without train nodes: agree on the maximum once per epoch and let the
        # ranks that run out join the remaining steps' collectives with zero gradients (`_null_step`).
        loader = [b for b in cluster_loader if _has_train_nodes(b)]
        n_null = sync.max_steps(len(loader), torch.device(device)) - len(loader)
    for batch, batch_after in _with_lookahead(loader, graphs is not None and mode == 'learned'):
        if not _has_train_nodes(batch):
            continue
        ops.new_memo_scope()                    # per-step memos (x W^T shared by the learned and the random forward) die here
        total_update += 1
        # training_hybrid.py:52-53 zeroes both optimisers' gradients at the top of every step.  A REPLAYED step neither accumulates into
        # .grad nor reads it: its captured backward writes static buffers, which h.backward() then attaches to every parameter (None for
        # the ones without a gradient) -- and two torch zero_grad() calls are ~50 us of host time, more than the host has to spare beside
        # an unsampled step's 140 us of GPU work.  So in graph mode the zeroing happens only for steps that run eagerly.
        if graphs is None or mode != 'learned' or _ALWAYS_ZERO_GRAD:
            optimizer_edge_prob.zero_grad()
            optimizer_gnn.zero_grad()

        if mode == 'learned':
            batch = batch.to(device)
            # `h` (stepgraph.py, opt-in): the same segments replayed from captured HIP graphs instead of launched one by one;
            # naming the following batch lets it issue that partition's parameter-independent prefix ahead
            if graphs is not None:
                nxt = batch_after.to(device) if batch_after is not None and _has_train_nodes(batch_after) else None
                h = graphs.forward(batch, nxt)
                if not getattr(h, "replayed", False) and not _ALWAYS_ZERO_GRAD:
                    optimizer_edge_prob.zero_grad()
                    optimizer_gnn.zero_grad()
            else:
                h = None
            eager_opt = h is None or not h.opt_in_graph       # capturable optimisers are stepped inside the backward graph
            # N > 1 with FusedAdam: a replayed step all-reduces the gradient bucket and replays the optimiser graph inside
            # h.backward(); the "any rank learned" word travels in that bucket and is read on the device
            fused_dp = h is not None and graphs.dp and h.dp_in_handle
            global_gate = bool(getattr(args, "sgs_dp_global_gate", False))
            esync = sync if not fused_dp else None            # the eager collectives below are for every other case
            sampled = h.sampled if h is not None else batch.edge_index.shape[1] > q
            if sampled and h is None:
                ops.get_pairs(batch.edge_index, batch.x.shape[0], build=True)      # once per partition (cached): mates for the paired scorer forward
            if sampled:
                st = sampled_forward(pipeline, args, model, batch, q, use_checkpoint, noise) if h is None else None
                temperature = max(args.t_min, args.t_init - epoch * ((args.t_init - args.t_min) / max_epoch))   # returned, never used by the sampler

                update_edge_mlp = True
                counts = None
                any_learned = False
                if args.conditional and fused_dp and global_gate:
                    sync.gate_sum(h.cbuf[0:4])                                     # one gate for the union of the ranks' batches
                    counts = h.cbuf.tolist()
                    counts = [counts[0:2], counts[2:4]]
                    update_edge_mlp = counts[0][0] > counts[1][0]                  # micro-F1 over the union: same denominator on both sides
                elif args.conditional and h is not None and esync is None:
                    counts = h.gate_counts() + [0]                                 # replay: polled from pinned host memory
                    counts = [counts[0:2], counts[2:4]]
                    update_edge_mlp = counts[0][0] > counts[1][0]
                elif args.conditional:
                    cbuf = st.cbuf if h is None else h.cbuf
                    if esync is not None:           # N > 1: does ANY rank's gate choose "learned"? (device-side, no extra sync)
                        cbuf[4:5] = esync.any_learned((cbuf[0:1] > cbuf[2:3]).to(torch.int32))
                    counts = cbuf.tolist()                                         # the step's one host read-back
                    any_learned = esync is not None and counts[4] > 0
                    counts = [counts[0:2], counts[2:4]]
                    # learned_f1 > random_f1 with f1 = correct / n_train on both sides (utils.py:163-169)
                    update_edge_mlp = counts[0][0] > counts[1][0]

                if esync is not None and not args.conditional:
                    esync.any_learned(torch.ones(1, dtype=torch.int32, device=batch.x.device))   # keep the collective in lock-step
                if update_edge_mlp:
                    condtional_update += 1
                    if h is None:
                        loss = learned_loss(args, criterion, st, batch)
                        _backward(model, loss)
                    else:
                        with segment(model, "backward"):
                            loss = h.backward(True)
                    if esync is not None:
                        esync.sync()
                    if eager_opt:
                        optimizer_edge_prob.step()
                        optimizer_gnn.step()
                else:
                    if h is None:
                        loss = _ce(criterion, st.random_out, batch)
                        _backward(model, loss)
                    else:
                        with segment(model, "backward"):
                            loss = h.backward(False)
                    if esync is not None:
                        esync.sync(all_random=not any_learned)
                        if any_learned:
                            optimizer_edge_prob.step()      # another rank's gate chose "learned"
                    if eager_opt:
                        optimizer_gnn.step()

                if trace is not None:
                    trace.update(rsei=st.rsei, prior_sample=st.rs, edge_probs_full=st.edge_probs_full.detach(), sample=st.smp,
                                 w=st.edge_probs_for_loss.detach(), learned_out=st.learned_out.detach(),
                                 random_out=None if st.random_out is None else st.random_out.detach(), counts=counts,
                                 update_edge_mlp=update_edge_mlp, loss=loss.detach())
            else:
                if fused_dp and global_gate and args.conditional:
                    sync.gate_sum(torch.zeros(4, dtype=torch.int32, device=batch.x.device))   # keep the gate collective in lock-step
                if h is None:
                    out = model(batch, batch.edge_index)
                    loss = _ce(criterion, out, batch)
                    _backward(model, loss)
                else:
                    with segment(model, "backward"):
                        loss = h.backward(None)
                if esync is not None:
                    # every rank is in lock-step on the partition stream, so ranks whose partition is small
                    # (no sampling, no gate) still join the flag all-reduce and the gradient all-reduce
                    flag = esync.any_learned(torch.zeros(1, dtype=torch.int32, device=batch.x.device))
                    some_learned = int(flag.item()) > 0
                    esync.sync(all_random=not some_learned)
                    if some_learned:
                        optimizer_edge_prob.step()
                if eager_opt:
                    optimizer_gnn.step()

        elif mode == 'random':
            batch = batch.to(device)
            if batch.edge_index.shape[1] > q:
                out = model(batch, random_edge_sampling(batch.edge_index, q=q))
            else:
                out = model(batch, batch.edge_index)
            loss = _ce(criterion, out, batch)
            _backward(model, loss)
            if sync is not None:
                sync.sync()
            optimizer.step()

        elif mode == 'edge':
            batch = batch.to(device)
            if batch.edge_index.shape[1] > q:
                out = model(batch, draw_prior(batch.prob, batch.edge_index, q).edge_index)
            else:
                out = model(batch, batch.edge_index)
            loss = _ce(criterion, out, batch)
            _backward(model, loss)
            if sync is not None:
                sync.sync()
            optimizer.step()

        elif mode == 'full':
            batch = batch.to(device)
            out = model(batch, batch.edge_index)
            loss = _ce(criterion, out, batch)
            _backward(model, loss)
            if sync is not None:
                sync.sync()
            optimizer.step()

        else:
            raise ValueError("Invalid mode. Choose 'learned', 'random', or 'full'.")

        if h is not None and mode == 'learned' and h.loss_on_device:
            pass                            # replayed graphs add their loss to graphs.loss_sum on the device
        else:
            total_loss = loss.detach().clone() if total_loss is None else total_loss + loss.detach()

    for _ in range(n_null):
        _null_step(args, mode, model, optimizer_gnn, optimizer_edge_prob, optimizer, sync, graphs, torch.device(device))

    loss_total = float(total_loss) if total_loss is not None else 0.0
    if graphs is not None:
        loss_total += float(graphs.loss_sum)
    mean_loss = loss_total / len(cluster_loader)
    return mean_loss, temperature, condtional_update, total_update
