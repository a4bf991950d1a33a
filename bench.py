#!/usr/bin/env python3
"""Headline benchmark: sampled edges/s + training steps/s of the SGS-GNN hybrid pipeline on a
Reddit-like METIS-partition stream (20 % of edges kept), on N MI355X of one node.

    python bench.py --gpus 1 --steps 230 --warmup 5
    python bench.py --gpus N --steps K --warmup W          # WORLD_SIZE unset: starts its own N ranks (one fresh process per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one iteration of the reference's hot loop (training_hybrid.train's loop body:
prior draw -> EdgeProbGCN scores for every edge -> learned draw -> weighted 2-layer GCN ->
second GCN on the random subgraph -> F1 gate -> CE + reg1 + reg2 -> backward -> Adam steps) on
one partition batch already resident in HBM.  Workload = SURVEY.md section 8d "S3": the 230-partition
stream of the reference's Reddit run (main.py:41-67: ClusterData / ClusterLoader(batch_size=1, shuffle=True)) --
partitions of ~1013 nodes, F=602, C=41, H=256, intra-partition edges in [60k, 500k] with 52 % above
q = 100 000 (119 of 230 partitions, logs/pipeline_hybrid.log:8), visited in a shuffled order, dropout 0.3,
conditional gate on, both regularisers on, fp32.  Data are synthetic (no network for Reddit).  The timed region is
K consecutive steps of that shuffled stream after W untimed ones; with one GPU the line also carries whole-epoch
timings (`epochs`), per-branch step times (`branch_ms`) and the one-time capture cost (`capture_s`).

Prints ONE JSON line (rank 0).  `value` = learned-sampled edges per second over all ranks
(q per step whose partition has more than q edges; the prior-only draw is not counted).

Before the timed window one whole epoch of the stream runs untimed (`--settle-epochs`, 0.14 s): the gate moves from "random wins" at
initialisation to "learned wins" within the first epoch and a learned-branch step costs twice a random-branch one, so a window
taken right after initialisation flatters the number; `value` is the steady-state figure (`learned_fraction_in_window`).

Other BASELINE.json configurations (one JSON line each, same metric, `config.workload` names them):
    python bench.py --config S2      # CitationFull-Cora-like full graph (N=19 793, F=8 710, E=126 842, q=25 368)
    python bench.py --config S4      # arxiv-year-like partitions, --GNN GAT, straight_through
    python bench.py --config S5 --gpus N   # ONE full-Reddit-size graph (N=232 965, E=114.6 M, q=22.9 M), edge-partitioned over N
                                           # ranks: STRONG scaling (N=1 = plain train()); all-reduce form and node-block form
The default (S3) line carries the S5 record as a sub-object (`s5`; `--s5 0` skips it), so one driver line per N holds both the
weak-scaling partition stream and the strong-scaling edge-sharded step.
"""
import argparse
import contextlib
import io
import json
import os
import random
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

Q = 100_000
N_NODES, NFEAT, NCLS, HID = 1013, 602, 41, 256
F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense fp32 matrix peak (= vector peak)
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 matrix peak
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E
# forward variant 4 evaluates every fp32 product as SIX bf16 MFMA products over exact 3-way operand splits (fp32-faithful,
# csrc/edge_score.hip): its matrix pipe is the bf16 one and it executes 6x the algorithmic flops, so the kernel's roofline
# is the bf16 peak / 6 in algorithmic fp32 flops.
BF16X6_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 6.0
METRIC = "sampled edges/sec + training steps/sec, Reddit hybrid 20% sparsity"


def make_args(device, **kw):
    a = argparse.Namespace(
        device=device, mode="learned", pipeline="hybrid", edge_mlp_type="GCN", conditional=True, sparse_edge_mlp=True,
        t_init=0.7, t_min=0.5, degree_bias_coef=0.3, reg1=True, reg2=True, regularizer1_coef=1.0, consist_reg_coef=0.5,
        hybrid_checkpoint=True, drop_rate=0.3, lr=1e-3)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def build_model(S, device, fused=True, nfeat=NFEAT, hid=HID, ncls=NCLS, gnn="GCN"):
    """Same two Adam optimisers over the same (overlapping) parameter sets as main.py:100-109,122; `fused` selects
    sgs_gnn_amd.FusedAdam (one launch per group, torch.optim.Adam's update rule and state layout, capturable) so that a
    replayed step includes its optimiser steps; 0 = torch.optim.Adam (foreach), stepped eagerly after each replay."""
    torch.manual_seed(42)
    if gnn == "GAT":
        m = S.GATModel(nfeat, hid, ncls, dropout_prob=0.3, edge_mlp_type="GCN").to(device)
    else:
        m = S.GNNModel(nfeat, hid, ncls, dropout_prob=0.3, edge_mlp_type="GCN").to(device)
    Adam = S.FusedAdam if fused else torch.optim.Adam
    opt_gnn = Adam([p for n, p in m.named_parameters() if "gcn" in n or "GAT" in n], lr=1e-3)      # main.py:100-109
    opt_edge = Adam([p for n, p in m.named_parameters() if "edge_prob_mlp" in n], lr=1e-3)         # main.py:122
    opt_all = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=5e-4)                          # main.py:123
    return m, opt_gnn, opt_edge, opt_all


def _hip_time(fn, reps, warm=3):
    """Average duration of fn() by HIP events on torch's current stream (the stream the ABI launches on)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def _pmc_traffic(E):
    """HBM bytes per launch of the scorer forward from the separate rocprofv3 --pmc passes on this very kernel and shape
    (profiles/r0*_scorer_pmc.json: 2 x FETCH_SIZE + WRITE_SIZE, gfx950 correction applied); None if the resident shape
    differs from the profiled one."""
    names = ("r03_scorer_paired_pmc.json", "r02_scorer_paired_pmc.json") if E < 0 else ("r02_scorer_pmc.json", "r01_scorer_pmc.json")      # E < 0: the paired entry point
    for name in names:
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", name)))
            if f"E={abs(E)}," in pmc["kernel"]:
                return pmc["hbm_traffic_bytes_per_launch"]
        except Exception:
            pass
    return None


def kernel_roofline(S, model, batch, reps, n_nodes=N_NODES, hid=HID, paired=False):
    """Live HIP-event timing of the dominant kernel (the fused MFMA edge scorer, forward over all E
    candidate edges of a partition) on the stream it is launched on (torch's current stream).
    `paired`: the entry point the training step uses on an undirected graph stored both ways (sgs_edge_score_fwd_paired: the
    canonical edge of every (s -> d), (d -> s) pair runs the contraction, both scores come out of its epilogue)."""
    ops = S.ops
    E = batch.edge_index.shape[1]
    H = hid
    sc = model.edge_prob_mlp
    codes = torch.relu(torch.randn(n_nodes, H, device=batch.x.device))
    L = S._lib.lib()
    U = (codes @ sc.fc1.weight[:, H:].t()).contiguous()
    out = torch.empty(E, dtype=torch.float32, device=codes.device)
    ws = ops.workspace(L.sgs_edge_score_workspace_bytes(n_nodes, H, E), codes.device)
    W1, b1, w2, b2 = sc.fc1.weight.detach().contiguous(), sc.fc1.bias.detach(), sc.fc2.weight.detach().reshape(-1).contiguous(), sc.fc2.bias.detach()

    M = E
    if paired:
        canon, mate = ops.get_pairs(batch.edge_index, n_nodes, build=True)
        M = int(canon.numel())

    maskbits = torch.empty(E, H // 32, dtype=torch.int32, device=codes.device) if paired else None

    def launch():
        if paired:
            # what a TRAINING step's forward runs (ops._EdgeScore.forward with gradients enabled): the paired loop that also keeps the
            # ReLU x dropout mask of every scored edge for the backward
            S._lib.check(L.sgs_edge_score_fwd_mask(codes.data_ptr(), U.data_ptr(), n_nodes, H, batch.edge_index.data_ptr(), E, 0, canon.data_ptr(), M,
                                                   mate.data_ptr(), W1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), 0.3, 1, 2, out.data_ptr(),
                                                   maskbits.data_ptr(), ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream),
                         "edge_score_fwd_mask")
        else:
            S._lib.check(L.sgs_edge_score_fwd(codes.data_ptr(), U.data_ptr(), n_nodes, H, batch.edge_index.data_ptr(), E, 0, W1.data_ptr(),
                                              b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), 0.3, 1, 2, out.data_ptr(), ws.data_ptr(),
                                              ws.numel(), torch.cuda.current_stream().cuda_stream), "edge_score_fwd")
    ms = _hip_time(launch, reps)
    flops = E * (2.0 * H * H + 2.0 * H)          # algorithmic flops per launch after the W1 split: per CANDIDATE edge
    executed = M * 2.0 * H * H + E * 2.0 * H     # what the launch actually issues (the H x H contraction for the M canonical edges only)
    achieved = flops / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": "sgs_edge_score_fwd", "achieved": round(achieved, 3),
            "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / F32_MFMA_PEAK_TFLOPS, 4),
            "traffic": _pmc_traffic(E) if not paired else _pmc_traffic(-E), "edges_per_launch": E, "canonical_edges_per_launch": M,
            "ms_per_launch": round(ms, 4), "flops_per_edge": 2 * H * H + 2 * H,
            "executed_tflops": round(executed / (ms * 1e-3) / 1e12, 3)}


def scorer_roofline(S, model, big, score_variant, n_nodes=N_NODES, hid=HID, alts=True):
    L = S._lib.lib()
    names = {0: "lds_tiled", 1: "stream_32_edge_wave_tile", 2: "weight_stationary_persistent", 3: "stream_64_edge_wave_tile",
             4: "bf16x6_split_on_bf16_mfma", 5: "bf16x6_paired"}
    auto = 4 if hid % 128 == 0 else 3
    used = score_variant if score_variant >= 0 else auto           # automatic choice at this E (>= 65 536 edges)
    paired = score_variant < 0 and bool(L.sgs_edge_score_paired_supported(hid))      # what the steps run on an undirected graph (ops.edge_score)
    alt = {}
    if alts:
        for v, name in names.items():                                  # in-process A/B of the scorer forward kernels
            if v == 5 or (v == used and not paired):
                continue
            L.sgs_edge_score_set_variant(v)
            r_ = kernel_roofline(S, model, big, reps=20, n_nodes=n_nodes, hid=hid)
            alt[name] = {"achieved": r_["achieved"], "ms_per_launch": r_["ms_per_launch"]}
    L.sgs_edge_score_set_variant(score_variant)
    roof = kernel_roofline(S, model, big, reps=20, n_nodes=n_nodes, hid=hid, paired=paired)       # the entry point the timed steps use
    if paired:
        used = 5
        roof["kernel"] = ("sgs_edge_score_fwd_mask (the training forward: paired bf16x6 loop, MODE 3 -- canonical edges run the contraction, mates "
                          "ride along -- that also keeps the ReLU x dropout mask of every scored edge for the backward)")
        roof["peak"] = round(BF16X6_PEAK_TFLOPS, 1)
        roof["frac"] = round(roof["achieved"] / BF16X6_PEAK_TFLOPS, 4)
        roof["peak_note"] = ("achieved = ALGORITHMIC fp32 flops (2 H^2 + 2 H per CANDIDATE edge, SURVEY.md 8d) / time; peak = dense bf16 MFMA peak 2500 / 6 "
                             "(six bf16 MFMA products per fp32 product, exact 3-way splits).  The launch ISSUES the H x H contraction only for the "
                             "canonical edge of every (s->d, d->s) pair -- `executed_tflops`, `executed_frac` price that work")
        roof["executed_frac"] = round(roof["executed_tflops"] / BF16X6_PEAK_TFLOPS, 4)
        roof["bf16_mfma_tflops_executed"] = round(6 * roof["executed_tflops"], 1)
        roof["vs_fp32_mfma_peak"] = round(roof["achieved"] / F32_MFMA_PEAK_TFLOPS, 4)
        if alts:
            roof["alt_variants"] = alt
        return roof
    roof["kernel"] = f"sgs_edge_score_fwd, forward variant {used} ({names[used]})"
    if used == 4:
        roof["peak"] = round(BF16X6_PEAK_TFLOPS, 1)
        roof["frac"] = round(roof["achieved"] / BF16X6_PEAK_TFLOPS, 4)
        roof["peak_note"] = ("algorithmic fp32 flops; the kernel runs 6 bf16 MFMA products per fp32 product (exact 3-way splits, "
                             "fp32-faithful), so peak = dense bf16 MFMA peak 2500 / 6")
        roof["bf16_mfma_tflops_executed"] = round(6 * roof["achieved"], 1)
        roof["vs_fp32_mfma_peak"] = round(roof["achieved"] / F32_MFMA_PEAK_TFLOPS, 4)
    if alts:
        roof["alt_variants"] = alt
    return roof


def cpu_baseline(batch_cpu, nfeat=NFEAT, hid=HID, ncls=NCLS, q=Q, warm=3, timed=10):
    """The oracle's hybrid step (the reference's op sequence on CPU: materialised [E,2H] scorer,
    torch topk draw, gather->mul->index_add GCN, losses, autograd, two Adam steps) on the host
    cores, on a bounded sample of the same workload.  Protocol of BASELINE.md section 2: `warm` untimed
    + `timed` timed steps, median."""
    from oracle import sgs_oracle as O
    threads = min(os.cpu_count(), 16)          # the GPU box's CPU share for one GPU
    torch.set_num_threads(threads)
    n = batch_cpu.x.shape[0]
    P = {k: v.requires_grad_(True) for k, v in O.init_params(nfeat, hid, ncls, "GCN", seed=1).items()}
    cfg = O.StepConfig(pipeline="hybrid", scorer="GCN", q=q, conditional=True, drop_rate=0.3)
    b = dict(x=batch_cpu.x, edge_index=batch_cpu.edge_index, y=batch_cpu.y, train_mask=batch_cpu.train_mask, prob=batch_cpu.prob)
    E = batch_cpu.edge_index.shape[1]
    g = torch.Generator().manual_seed(0)
    st_e, st_g = {}, {}

    def one():
        nz = O.StepNoise(prior_noise=torch.empty(E).exponential_(1, generator=g), sample_noise=torch.empty(E).exponential_(1, generator=g))
        nz.masks_pass1 = O.Masks(enc_hidden=torch.rand(n, hid, generator=g) > 0.3, score_hidden=torch.rand(E, hid, generator=g) > 0.3)
        nz.gnn_keep_learned = torch.rand(n, hid, generator=g) > 0.3
        nz.gnn_keep_random = torch.rand(n, hid, generator=g) > 0.3
        R = O.learned_step_forward(P, b, cfg, nz)
        for p_ in P.values():
            p_.grad = None
        R["loss"].backward()
        grads = {k: v.grad for k, v in P.items()}
        with torch.no_grad():
            if R["update_edge_mlp"]:
                O.adam_step({k: v for k, v in P.items() if "edge_prob_mlp" in k}, grads, st_e)
            O.adam_step({k: v for k, v in P.items() if "gcn" in k}, grads, st_g)
    for _ in range(warm):
        one()
    ts = []
    for _ in range(timed):
        t0 = time.perf_counter()
        one()
        ts.append(time.perf_counter() - t0)
    med = statistics.median(ts)
    return {"value": round(q / med, 1), "unit": "sampled edges/s", "steps_per_s": round(1.0 / med, 4), "cores": threads,
            "kind": "port", "sample": f"median of {timed} hybrid steps after {warm} warm-up steps (BASELINE.md protocol) on one synthetic "
                                      f"partition, n={n}, E={E}, q={q}, F={nfeat}, H={hid}, fp32, torch {torch.__version__} CPU; "
                                      f"min {min(ts):.3f} s, max {max(ts):.3f} s per step"}


def pool_indices(sizes, rank, world, per_rank):
    """Which partitions of the common stream rank `rank` holds.  Data-parallel steps end in a gradient all-reduce, so a step lasts
    as long as its slowest rank: the stream is dealt out BY SIZE (sorted by edge count, rank r takes every world-th one), so that
    the partitions the ranks process in the same step have adjacent sizes -- in particular all sampled or all unsampled -- and the
    step order is then shuffled with a permutation common to all ranks."""
    order = sorted(range(len(sizes)), key=lambda i: (sizes[i], i))
    mine = order[rank::world][:per_rank]
    perm = random.Random(1000).sample(range(len(mine)), len(mine))
    return [mine[i] for i in perm]


def make_pool(S, rank, world, per_rank, device):
    if world == 1:
        return S.reddit_partition_stream(num_parts=per_rank, seed=1000, nfeat=NFEAT, ncls=NCLS, n=N_NODES, q=Q, device=device)
    sizes = S.reddit_partition_sizes(per_rank * world, seed=1000, q=Q)
    idx = pool_indices(sizes, rank, world, per_rank)
    parts = S.reddit_partition_stream(num_parts=per_rank * world, seed=1000, nfeat=NFEAT, ncls=NCLS, n=N_NODES, q=Q, device=device,
                                      only=set(idx))
    return [parts[i] for i in idx]


def _quiet_train(S, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return S.train(*a, **k)


def _timed_train(S, world, *a, **k):
    """barrier + synchronize on both sides; returns (seconds, train()'s return tuple)."""
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ret = _quiet_train(S, *a, **k)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    return time.perf_counter() - t0, ret


def _reduce(world, device, dt, sampled):
    t = torch.tensor([dt, float(sampled)], dtype=torch.float64, device=device)
    if world > 1:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        return float(tmax[0]), float(tsum[1])
    return dt, float(sampled)


# --------------------------------------------------------------------------------------------- S3 (headline)
def run_s3(a, S, rank, world, device):
    model, opt_gnn, opt_edge, opt_all = build_model(S, device, fused=bool(a.fused_adam))
    crit = torch.nn.CrossEntropyLoss()
    args = make_args(device)
    t0 = time.perf_counter()
    pool = make_pool(S, rank, world, a.parts, device)       # resident in HBM before any timing
    torch.cuda.synchronize()
    pool_s = time.perf_counter() - t0
    P = len(pool)
    order = random.Random(7).sample(range(P), P)             # ClusterLoader(shuffle=True): one permutation, common to all ranks
    stream = lambda i: pool[order[i % P]]                    # noqa: E731
    train_args = (model, opt_gnn, opt_edge, opt_all, crit)

    capture_s, hbm0 = 0.0, torch.cuda.memory_allocated()
    if a.hipgraph:
        args.sgs_hipgraph = True
        args.sgs_dp_global_gate = world > 1      # N > 1: one gate per step over the union of the ranks' batches (dist.py)
        with contextlib.redirect_stdout(io.StringIO()):
            capture_s = S.prepare_step_graphs(args, model, opt_gnn, opt_edge, crit, pool, q=Q)
    graph_hbm = torch.cuda.memory_allocated() - hbm0

    warm = [stream(i) for i in range(a.warmup)]
    timed = [stream(a.warmup + i) for i in range(a.steps)]
    sampled = sum(Q for b in timed if b.edge_index.shape[1] > Q)
    # untimed: whole epochs of the stream until the gate mix has settled (module docstring); every rank runs the same number of steps
    settle = []
    for e in range(a.settle_epochs):
        so = random.Random(50 + e).sample(range(P), P)
        r_ = _quiet_train(S, args, 0, 10, *train_args, [pool[i] for i in so], q=Q, alternate_frequency=0)
        settle.append(r_[2])
    if warm:
        _quiet_train(S, args, 0, 10, *train_args, warm, q=Q, alternate_frequency=0)
    dt, ret = _timed_train(S, world, args, 1, 10, *train_args, timed, q=Q, alternate_frequency=0)
    dt_all, sampled_all = _reduce(world, device, dt, sampled)
    n_above = sum(1 for b in timed if b.edge_index.shape[1] > Q)

    rec = {
        "metric": METRIC,
        "value": round(sampled_all / dt_all, 1), "unit": "sampled edges/s",
        "steps_per_s": round(a.steps * world / dt_all, 3),
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt_all / a.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "dtype_note": "fp32 tensors, accumulators and results throughout; the scorer's fp32 contractions (forward, and the backward's "
                      "recompute where one is needed) run as six bf16 MFMA products over exact 3-way operand splits (error measured at the fp32-MFMA kernels' "
                      "level, see roofline.peak_note); the backward's dv.W1a and weight-gradient GEMM take dv as its exact 0/1 mask "
                      "(ReLU x dropout, kept by the training forward) times row / column factors, so one operand is a single exact bf16 piece: "
                      "three products; no recompute of the hidden layer in the backward",
        "data": "synthetic",
        "config": {"workload": f"Reddit-like METIS partition stream (S3): {P} partitions per GPU, shuffled, n=1013 F=602 C=41 H=256, E_b in "
                               "[60k,500k] (52% above q), q=100000, hybrid pipeline, EdgeProbGCN scorer, conditional gate, reg1+reg2, "
                               "dropout 0.3, Adam x2; partitions resident in HBM (no per-batch collation or H2D copy in the timed region, "
                               "unlike training_hybrid.py:42)",
                   "partitions": P, "partitions_above_q_in_window": n_above,
                   "parallelism": f"dp{world} (partition-sharded by size across ranks, global gate, 1 flat gradient all-reduce/step)" if world > 1 else "single",
                   "hipgraph_replay": bool(getattr(args, "sgs_hipgraph", False)),
                   "adam": "sgs_gnn_amd.FusedAdam (in-graph)" if a.fused_adam else "torch.optim.Adam (foreach, eager)"},
        "mean_loss": round(ret[0], 5), "conditional_updates": ret[2], "total_updates": ret[3],
        "learned_fraction_in_window": round(ret[2] / max(n_above, 1), 3),
        "settle_epochs": a.settle_epochs, "settle_learned_steps": settle,
        "capture_s": round(capture_s, 3), "graph_hbm_GiB": round(graph_hbm / 2**30, 3), "pool_build_s": round(pool_s, 2),
    }
    if world > 1:
        rec["collective_backend"] = dist.get_backend()
        rec["collective_ranks"] = dist.get_world_size()

    if rank == 0 and world == 1:
        # ---- whole epochs (the reference's timed region is a whole train() over all partitions, main.py:147-168)
        epochs = []
        for e in range(a.epochs):
            eo = random.Random(100 + e).sample(range(P), P)
            ep_batches = [pool[i] for i in eo]
            dte, re_ = _timed_train(S, world, args, 2 + e, 10, *train_args, ep_batches, q=Q, alternate_frequency=0)
            n_s = sum(1 for b in ep_batches if b.edge_index.shape[1] > Q)
            epochs.append({"seconds": round(dte, 4), "steps_per_s": round(P / dte, 2), "sampled_edges_per_s": round(n_s * Q / dte, 1),
                           "sampled_steps": n_s, "learned_steps": re_[2], "mean_loss": round(re_[0], 4)})
        if epochs:
            rec["epochs"] = epochs
            rec["epoch1_incl_capture_s"] = round(capture_s + epochs[0]["seconds"], 4)
        # ---- per-branch step times: one step per train() call with a device synchronisation around it (no look-ahead overlap)
        if a.diag_steps > 0:
            br = {"learned": [], "random": [], "unsampled": []}
            for i in range(a.diag_steps):
                b = stream(1000 + i)
                dts, r_ = _timed_train(S, world, args, 5, 10, *train_args, [b], q=Q, alternate_frequency=0)
                kind = "unsampled" if b.edge_index.shape[1] <= Q else ("learned" if r_[2] else "random")
                br[kind].append(dts * 1e3)
            rec["branch_ms"] = {k: (round(statistics.mean(v), 4) if v else None) for k, v in br.items()}
            rec["branch_ms"]["n"] = {k: len(v) for k, v in br.items()}
            rec["branch_ms"]["note"] = "one step per train() call, synchronised: includes the per-call host overhead, no prefetch overlap"
        rec["hbm_reserved_GiB"] = round(torch.cuda.memory_reserved() / 2**30, 2)
        big = max(pool, key=lambda b: b.edge_index.shape[1])
        rec["roofline"] = scorer_roofline(S, model, big, a.score_variant, alts=bool(a.alts))
        if not a.no_cpu_baseline:
            above = [b for b in pool if b.edge_index.shape[1] > Q]
            cpu_b = min(above, key=lambda b: b.edge_index.shape[1]) if above else big      # smallest sampled partition: bounded CPU time
            rec["cpu_baseline"] = cpu_baseline(cpu_b.to("cpu"))
    elif rank == 0:
        big = max(pool, key=lambda b: b.edge_index.shape[1])
        rec["roofline"] = scorer_roofline(S, model, big, a.score_variant, alts=False)
    return rec


# --------------------------------------------------------------------------------------------- S2: CoraFull-like full graph
def corafull_like(S, device, seed=5):
    """CitationFull-Cora shapes (datasets.py:51-54; logs/log_macro.txt:37): N=19 793, F=8 710, 126 842 directed edges, 70 classes;
    features: 0.7 % dense non-negative bag-of-words rows, L1-normalised; masks 0.2/0.4/0.4 (datasets.py:201)."""
    N, F_, E, C = 19_793, 8_710, 126_842, 70
    b = S.synthetic_graph(N, E, 8, C, seed=seed, train_frac=0.2, power=0.5, device=device)
    g = torch.Generator(device=device).manual_seed(seed)
    x = (torch.rand(N, F_, device=device, generator=g) < 0.007).float() * torch.rand(N, F_, device=device, generator=g)
    x[torch.arange(N, device=device), b.y * 100 % F_] += 0.5
    b.x = x / x.sum(1, keepdim=True).clamp_min(1e-12)
    return b


def run_s2(a, S, device):
    Fin, C = 8_710, 70
    b = corafull_like(S, device)
    E = b.edge_index.shape[1]
    q = int(E * 0.2)                                        # main.py:54: un-partitioned graph
    model, opt_gnn, opt_edge, opt_all = build_model(S, device, fused=bool(a.fused_adam), nfeat=Fin, ncls=C)
    crit = torch.nn.CrossEntropyLoss()
    args = make_args(device)
    if a.hipgraph:
        args.sgs_hipgraph = True
    loader = [b]                                            # main.py:67: cluster_loader = [data]
    cap = 0.0
    if a.hipgraph:
        with contextlib.redirect_stdout(io.StringIO()):
            cap = S.prepare_step_graphs(args, model, opt_gnn, opt_edge, crit, loader, q=q)
    for ep in range(a.warmup):
        _quiet_train(S, args, ep, 10, model, opt_gnn, opt_edge, opt_all, crit, loader, q=q)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    learned = 0
    for ep in range(a.steps):
        r = _quiet_train(S, args, ep, 10, model, opt_gnn, opt_edge, opt_all, crit, loader, q=q)
        learned += r[2]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # dominant kernel here (profiles/r03_s2_kernel_stats.csv): the node-level product x W^T of the first GCN layers over the NON-ZEROS of the
    # 0.7 %-dense bag-of-words rows (ops.FeatCSR -> sgs_spmm_csr gathering rows of W^T), four launches per step; HBM-bound gather, priced per
    # SURVEY.md 8d's SpMM row: per nnz 4 B column + 4 B value + 4 H B gathered row; 4 H N B output; 8 (N + 1) B row pointers
    ops = S.ops
    W = model.gcn1.lin.weight.detach()
    fc = ops.feature_csr(b.x, build=True)
    if fc is None:
        raise SystemExit("S2: the feature matrix did not take the sparse path")
    with torch.no_grad():
        ms = _hip_time(lambda: ops._x_wt(b.x, W), 20)                     # (includes the [F, H] transpose copy of W, as every step does)
        Wt = W.t().contiguous()
        ms_k = _hip_time(lambda: ops._spmm(Wt, fc.ptr, fc.col, fc.val, None, None, ops.ACT_NONE, 0.0, 0, 0, fc.N, HID, fc.nnz), 20)
        ms_dense = _hip_time(lambda: torch.mm(b.x, W.t()), 5)
    Nn = b.x.shape[0]
    alg = fc.nnz * (8 + 4 * HID) + 4 * HID * Nn + 8 * (Nn + 1)
    compulsory = fc.nnz * 8 + 4 * HID * Fin + 4 * HID * Nn + 8 * (Nn + 1)   # W^T read once instead of once per non-zero
    rec = {"metric": METRIC, "value": round(q * a.steps / dt, 1), "unit": "sampled edges/s", "steps_per_s": round(a.steps / dt, 3), "n_gpus": 1,
           "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"CitationFull-Cora-like full graph (S2): N={b.x.shape[0]} F={Fin} C={C} H={HID} E={E} q={q}, one batch per "
                                  "epoch (main.py:67), hybrid pipeline, EdgeProbGCN scorer, conditional gate, reg1+reg2, dropout 0.3, Adam x2",
                      "hipgraph_replay": bool(a.hipgraph)},
           "conditional_updates": learned, "capture_s": round(cap, 3),
           "roofline": {"bound": "hbm", "kernel": f"sparse-feature node product x W^T over nnz(x) = {fc.nnz} (sgs_spmm_csr, D = {HID}; "
                                                  "spmm_csr_rowblock<4, 4>): the step's largest kernel, 4 launches per step",
                        "achieved": round(compulsory / (ms_k * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(compulsory / (ms_k * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None, "ms_per_launch": round(ms_k, 4),
                        "algorithmic_bytes": compulsory, "uncached_gather_bytes": alg,
                        "uncached_gather_GBps": round(alg / (ms_k * 1e-3) / 1e9, 1),
                        "ms_with_weight_transpose": round(ms, 4), "ms_dense_library_gemm": round(ms_dense, 4),
                        "note": "achieved = SURVEY 8d's COMPULSORY figure (CSR of x, W^T and the output once each): 39 MB per launch, so the "
                                "kernel is latency-, not HBM-bound at this size.  uncached_gather_GBps counts the gathered W^T row once per "
                                "non-zero (8d's upper figure; the 8.9 MB table is L2 / Infinity-Cache resident: gather bandwidth, above the "
                                "HBM peak).  The dense [N, F] x [F, H] library GEMM this replaces: ms_dense_library_gemm"}}
    if not a.no_cpu_baseline:
        rec["cpu_baseline"] = cpu_baseline(b.to("cpu"), nfeat=Fin, ncls=C, q=q, warm=1, timed=3)
    return rec


# --------------------------------------------------------------------------------------------- S4: arxiv-year-like, GAT, straight-through
def run_s4(a, S, device):
    Fin, C, n, Eb, parts = 128, 5, 33_869, 463_000, 5
    q = Q                                                   # METIS-partitioned: q = threshold * sample_perc (main.py:50)
    pool = [S.synthetic_graph(n, Eb, Fin, C, seed=300 + i, train_frac=0.2, power=0.6, device=device) for i in range(parts)]
    model, opt_gnn, opt_edge, opt_all = build_model(S, device, fused=bool(a.fused_adam), nfeat=Fin, ncls=C, gnn="GAT")
    crit = torch.nn.CrossEntropyLoss()
    args = make_args(device, pipeline="straight_through")
    cap = 0.0
    if a.hipgraph:
        args.sgs_hipgraph = True
        with contextlib.redirect_stdout(io.StringIO()):
            cap = S.prepare_step_graphs(args, model, opt_gnn, opt_edge, crit, pool, q=q)
    for ep in range(max(a.warmup // parts, 1)):
        _quiet_train(S, args, ep, 10, model, opt_gnn, opt_edge, opt_all, crit, pool, q=q)
    epochs = max(a.steps // parts, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    learned = 0
    for ep in range(epochs):
        learned += _quiet_train(S, args, ep, 10, model, opt_gnn, opt_edge, opt_all, crit, pool, q=q)[2]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = epochs * parts
    # dominant kernels (profiles/r03_s4_kernel_stats.csv): the straight-through pipeline scores and back-propagates ALL E_b candidate edges, so
    # the step is the scorer -- forward (paired bf16x6 loop, MFMA-bound) and the dense backward over every edge.  `roofline` prices the
    # forward, the largest MFMA kernel, exactly as the S3 line does; `gat_layer` keeps the K8 figure of round 2 (HBM-bound gather).
    ops = S.ops
    b0 = pool[0]
    roof = scorer_roofline(S, model, max(pool, key=lambda b: b.edge_index.shape[1]), a.score_variant, n_nodes=n, hid=HID, alts=False)
    smp = ops.sample_topq(ops.SAMPLE_PRIOR, b0.prob, None, 0.0, q, b0.edge_index, seed=1, stream_id=1, want_p=False)
    graph = ops.get_subgraph(b0.edge_index, n, smp)
    xl = torch.randn(n, HID, device=device)
    a_s, a_d = torch.randn(n, device=device), torch.randn(n, device=device)
    bias = torch.zeros(HID, device=device)
    with torch.no_grad():
        ms = _hip_time(lambda: ops.gat_aggregate(xl, a_s, a_d, bias, graph, 0.2, 0.3, 7, 16, ops.ACT_RELU_DROPOUT, 0.3, 7, 32), 20)
    nnz = q + n
    alg = nnz * (4 + 4 + 4 * HID + 8) + 4 * HID * n + 8 * (n + 1)           # col + weight + gathered row + 2 node scalars per nnz; output; row pointers
    roof["gat_layer"] = {"bound": "hbm", "kernel": "GAT layer forward over the drawn graph: sgs_gat_alpha_fwd + sgs_spmm_csr (D=256, nnz=q+n)",
                         "achieved": round(alg / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "ms_per_launch": round(ms, 4), "algorithmic_bytes": alg,
                         "note": "gathered rows counted per nnz (uncached upper figure, SURVEY 8d); the 35 MB feature table is L2 / "
                                 "Infinity-Cache resident"}
    rec = {"metric": METRIC, "value": round(q * steps / dt, 1), "unit": "sampled edges/s", "steps_per_s": round(steps / dt, 3), "n_gpus": 1,
           "steps": steps, "warmup": a.warmup, "ms_per_step": round(dt / steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"arxiv-year-like partitions (S4): {parts} partitions, n={n} F={Fin} C={C} H={HID} E_b~{pool[0].edge_index.shape[1]} "
                                  f"q={q}, --GNN GAT (heads 1), straight_through pipeline, EdgeProbGCN scorer, conditional gate, dropout 0.3, Adam x2",
                      "hipgraph_replay": bool(a.hipgraph)},
           "conditional_updates": learned, "capture_s": round(cap, 3),
           "roofline": roof}
    return rec


# --------------------------------------------------------------------------------------------- S5: one full-size graph, edge-partitioned
class _CollectiveMeter:
    """Counts the collectives of ONE step (calls, payload bytes per call) by wrapping torch.distributed's entry points."""

    def __init__(self):
        self.calls = []
        self._saved = {}

    def __enter__(self):
        def wrap(name, nbytes):
            fn = getattr(dist, name)
            self._saved[name] = fn

            def counted(*args, **kw):
                self.calls.append((name, nbytes(*args, **kw)))
                return fn(*args, **kw)
            setattr(dist, name, counted)
        wrap("all_reduce", lambda t, *a_, **k_: t.numel() * t.element_size())
        wrap("all_gather", lambda out, t, *a_, **k_: t.numel() * t.element_size() * len(out))
        wrap("reduce_scatter", lambda out, parts, *a_, **k_: sum(x.numel() * x.element_size() for x in parts))
        return self

    def __exit__(self, *exc):
        for name, fn in self._saved.items():
            setattr(dist, name, fn)
        return False

    def summary(self):
        by = {}
        for name, nb in self.calls:
            d = by.setdefault(name, {"calls": 0, "bytes": 0, "max_bytes": 0})
            d["calls"] += 1
            d["bytes"] += nb
            d["max_bytes"] = max(d["max_bytes"], nb)
        return {"calls": len(self.calls), "bytes": sum(nb for _, nb in self.calls), "by_op": by}


def run_s5(a, S, rank, world, device, steps=None, warmup=None):
    """BASELINE.json config 5: ONE Reddit-size graph (N = 232 965, F = 602, C = 41, E = 114.6 M candidate edges, q = 0.2 E = 22.9 M,
    main.py:54 un-partitioned), hybrid pipeline.  STRONG scaling: the same graph at every N.  N = 1: the plain single-GPU train()
    (eager launches -- a step is ~0.1-0.2 s of GPU work).  N > 1: the edge list is cut into N contiguous shards (sharded.py) and the
    step runs in both forms -- `train_step_sharded` (RCCL all-reduce of the [N, D] node embeddings, what north_star names) and
    `train_step_blocksharded` (node-block form: reduce-scatter forward / all-gather backward)."""
    from sgs_gnn_amd import sharded as sh
    steps = steps if steps is not None else a.steps
    warmup = warmup if warmup is not None else a.warmup
    N, F_, C, H = 232_965, NFEAT, NCLS, HID
    e_target = int(os.environ.get("SGS_BENCH_S5_EDGES", "114615892"))
    t0 = time.perf_counter()
    b = S.synthetic_graph(N, e_target, F_, C, seed=77, train_frac=0.66, power=0.35, device=device)     # same seed: every rank builds the same graph
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    E = b.edge_index.shape[1]
    q = int(E * 0.2)
    args = make_args(device)
    crit = torch.nn.CrossEntropyLoss()

    def fresh():
        S.fix_seeds(1)
        return build_model(S, device, fused=True)

    warm_ei = [None]

    def timed_steps(one_step):
        # the first warm-up step runs with the gate off, i.e. it takes the learned branch whatever the counts say: the caching allocator
        # grows by the learned backward's ~50 GB there (0.5 s of hipMalloc) and not inside the timed steps when the gate happens to pick
        # the random branch throughout the warm-up
        cond = args.conditional
        args.conditional = False
        try:
            one_step()
        finally:
            args.conditional = cond
        for _ in range(warmup):
            one_step()
        # set-up, like the CSR build above: the warm-up steps may all take the gate's random branch, and the FIRST learned-branch step then
        # pays one-time costs inside the timed region (growing the scratch arena by the sort workspace of the drawn subgraph, loading the
        # radix-sort code objects, the cached source-sortedness check of the edge list: ~1 s once, measured).  One draw + subgraph build here.
        with torch.no_grad():
            ei_w = b.edge_index if b.edge_index is not None else warm_ei[0]          # (sharded: this rank's slice)
            ew = ei_w.shape[1]
            smp = S.ops.sample_topq(S.ops.SAMPLE_LEARNED, torch.rand(ew, device=device), None, 0.0, max(min(q, ew - 1), 1), ei_w, seed=11, stream_id=3)
            S.ops.get_subgraph(ei_w, N, smp)
            del smp
        torch.cuda.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs = [one_step() for _ in range(steps)]
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt, _ = _reduce(world, device, time.perf_counter() - t0, 0)
        return dt, outs

    forms = {}
    if world == 1:
        model, og, oe, oa = fresh()
        ep = [0]

        def one():
            r = _quiet_train(S, args, ep[0], 10, model, og, oe, oa, crit, [b], q=q)
            ep[0] += 1
            return bool(r[2])
        dt, outs = timed_steps(one)
        forms["single_gpu_train"] = {"step_s": round(dt / steps, 5), "sampled_edges_per_s": round(q * steps / dt, 1), "learned_steps": sum(outs),
                                     "steps": steps}
    else:
        shard = sh.EdgeShard(b, rank, world)
        b.edge_index = b.prob = None                 # the shard holds this rank's slice; node data stay replicated
        warm_ei[0] = shard.edge_index
        torch.cuda.empty_cache()
        for name, fn in (("allreduce", sh.train_step_sharded), ("nodeblock", sh.train_step_blocksharded)):
            model, og, oe, oa = fresh()

            def one(fn=fn, model=model, og=og, oe=oe):
                return bool(fn(args, model, shard, og, oe, crit, q)["update_edge_mlp"])
            dt, outs = timed_steps(one)
            with _CollectiveMeter() as cm:
                learned = one()
            torch.cuda.synchronize()
            forms[name] = {"step_s": round(dt / steps, 5), "sampled_edges_per_s": round(q * steps / dt, 1), "learned_steps": sum(outs), "steps": steps,
                           "collectives_one_step": dict(cm.summary(), learned_branch=learned)}
    best = min(forms, key=lambda k: forms[k]["step_s"])
    rec = {"workload": f"full-Reddit-size graph (S5): N={N} F={F_} C={C} H={H} E={E} q={q}, hybrid pipeline, EdgeProbGCN scorer, conditional "
                       f"gate, reg1+reg2, dropout 0.3, FusedAdam x2; one step per epoch (main.py:67); synthetic",
           "scaling": "strong", "n_gpus": world, "steps": steps, "warmup": warmup, "best_form": best, "step_s": forms[best]["step_s"],
           "sampled_edges_per_s": forms[best]["sampled_edges_per_s"], "steps_per_s": round(1.0 / forms[best]["step_s"], 3), "forms": forms,
           "graph_build_s": round(build_s, 2), "peak_hbm_GiB": round(torch.cuda.max_memory_allocated() / 2**30, 1),
           "edges_per_rank": int(E // world)}
    if world > 1:
        rec["collective_backend"], rec["collective_ranks"] = dist.get_backend(), dist.get_world_size()
    del b
    torch.cuda.empty_cache()
    return rec


def s5_line(a, S, rank, world, device):
    """`--config S5` as its own bench line (same metric; strong scaling)."""
    r = run_s5(a, S, rank, world, device)
    best = r["forms"][r["best_form"]]
    rec = {"metric": METRIC, "value": r["sampled_edges_per_s"], "unit": "sampled edges/s", "steps_per_s": r["steps_per_s"], "n_gpus": world,
           "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(best["step_s"] * 1e3, 3), "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": r["workload"], "parallelism": (f"edge-partitioned over {world} ranks, form '{r['best_form']}'" if world > 1 else "single"),
                      "hipgraph_replay": False},
           "s5": r}
    return rec


def _free_port():
    import socket
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        return s_.getsockname()[1]


def spawn_ranks(n):
    """`python bench.py --gpus N` with WORLD_SIZE unset: start N fresh rank processes (this very command line, one per GPU) and relay
    rank 0's JSON line.  The parent makes NO GPU call, before or after (device_count() does not initialise the runtime on this
    image); nothing is re-exec'ed.  A failing rank takes the others down (exact PIDs) and the parent exits non-zero."""
    import subprocess
    import tempfile
    rehearse = os.environ.get("SGS_BENCH_REHEARSE") == "1"
    have = torch.cuda.device_count()
    if not rehearse and have < n:
        raise SystemExit(f"bench.py --gpus {n}: only {have} GPU(s) visible (SGS_BENCH_REHEARSE=1 runs all ranks on cuda:0 over gloo)")
    port = _free_port()
    procs, outs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        out = tempfile.TemporaryFile(mode="w+") if r == 0 else subprocess.DEVNULL
        outs.append(out)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=out))
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        for p in procs:
            if p.poll() not in (None, 0):
                failed = p.returncode
        time.sleep(0.05)
    if failed is None:
        failed = next((p.returncode for p in procs if p.returncode != 0), None)
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except Exception:
                p.kill()
    outs[0].seek(0)
    sys.stdout.write(outs[0].read())
    sys.stdout.flush()
    if failed is not None:
        raise SystemExit(failed if failed > 0 else 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=230, help="timed steps (default: one epoch of the 230-partition stream)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="S3", choices=["S3", "S2", "S4", "S5"])
    ap.add_argument("--settle-epochs", type=int, default=1, help="S3: untimed whole epochs before the timed window (the gate mix settles within the first)")
    ap.add_argument("--s5", type=int, default=1, help="S3: 1 (default) also runs config 5 (one full-size graph, edge-partitioned over the ranks; "
                    "strong scaling) and reports it as the `s5` sub-object; 0: skip")
    ap.add_argument("--s5-steps", type=int, default=4, help="timed steps per form of the `s5` sub-object (after 2 warm-up steps)")
    ap.add_argument("--parts", type=int, default=230, help="S3: partitions of the stream kept resident per rank")
    ap.add_argument("--epochs", type=int, default=2, help="S3, one GPU: additional whole-epoch timings")
    ap.add_argument("--diag-steps", type=int, default=48, help="S3, one GPU: synchronised single steps for the per-branch times")
    ap.add_argument("--pool", type=int, default=None, help="(deprecated alias of --parts)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--alts", type=int, default=1, help="1 (default): also time the other scorer forward kernels in-process (roofline.alt_variants); 0: skip")
    ap.add_argument("--fused-adam", type=int, default=1, help="1 (default): sgs_gnn_amd.FusedAdam (one launch per group, captured with the step); 0: torch.optim.Adam (foreach, eager)")
    ap.add_argument("--score-variant", type=int, default=-1, help="scorer forward kernel for the timed steps (benchmarking A/B; -1 = library default)")
    ap.add_argument("--hipgraph", type=int, default=1, help="1 (default): every step replayed from the HIP graphs captured once over static "
                    "slots (stepgraph.py); 0: eager launches")
    a = ap.parse_args()
    if a.pool is not None:
        a.parts = a.pool

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(a.gpus)             # the parent only launches and collects: no GPU call in this process
    if a.gpus != world:
        raise SystemExit(f"bench.py --gpus {a.gpus} inside a {world}-rank launch: pass the launcher's rank count")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the SGS hot path has no CPU fallback")
    # rehearsal on a one-GPU box (SGS_BENCH_REHEARSE=1): every rank on cuda:0 over gloo -- exercises the N > 1 code path
    # (pool dealing, graph-mode data parallel, collectives), says nothing about speed
    rehearse = os.environ.get("SGS_BENCH_REHEARSE") == "1"
    if rehearse:
        local = 0
    torch.cuda.set_device(local)
    device = f"cuda:{local}"
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearse:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device(device))

    import sgs_gnn_amd as S
    if a.score_variant >= 0:
        S._lib.lib().sgs_edge_score_set_variant(a.score_variant)
    S.fix_seeds(42 + rank)
    if a.config == "S3":
        rec = run_s3(a, S, rank, world, device)
        if a.s5:
            # after the S3 record is complete; a failure here (every rank raises the same way: same code, same shapes) must not cost the
            # S3 line, so it is reported inside the sub-object
            try:
                s5 = run_s5(a, S, rank, world, device, steps=a.s5_steps, warmup=2)
            except Exception as exc:                   # noqa: BLE001
                s5 = {"error": f"{type(exc).__name__}: {exc}"[:600]}
            if rank == 0:
                rec["s5"] = s5
    elif a.config == "S5":
        if a.steps == 230:
            a.steps, a.warmup = 6, 2
        rec = s5_line(a, S, rank, world, device)
    else:
        if world > 1:
            raise SystemExit("--config S2 / S4 are single-GPU lines")
        if a.steps == 230:
            a.steps = 20
        rec = run_s2(a, S, device) if a.config == "S2" else run_s4(a, S, device)
    if rank == 0:
        print(json.dumps(rec))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
