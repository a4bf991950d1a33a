#!/usr/bin/env python3
"""Headline benchmark: sampled edges/s + training steps/s of the SGS-GNN hybrid pipeline on a
Reddit-like METIS-partition stream (20 % of edges kept), on N MI355X of one node.

    python bench.py --gpus 1 --steps 60 --warmup 6
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one iteration of the reference's hot loop (training_hybrid.train's loop body:
prior draw -> EdgeProbGCN scores for every edge -> learned draw -> weighted 2-layer GCN ->
second GCN on the random subgraph -> F1 gate -> CE + reg1 + reg2 -> backward -> Adam steps) on
one partition batch already resident in HBM.  Workload = SURVEY.md section 8d "S3": partitions of
~1013 nodes, F=602, C=41, H=256, intra-partition edges in [60k, 500k] with 52 % above
q = 100 000 (the reference run: 119 of 230 partitions, logs/pipeline_hybrid.log:8), dropout 0.3,
conditional gate on, both regularisers on, fp32.  Data are synthetic (no network for Reddit).

Prints ONE JSON line (rank 0).  `value` = learned-sampled edges per second over all ranks
(q per step whose partition has more than q edges; the prior-only draw is not counted).
"""
import argparse
import json
import os
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
# forward variant 4 evaluates every fp32 product as SIX bf16 MFMA products over exact 3-way operand splits (fp32-faithful,
# csrc/edge_score.hip): its matrix pipe is the bf16 one and it executes 6x the algorithmic flops, so the kernel's roofline
# is the bf16 peak / 6 in algorithmic fp32 flops.
BF16X6_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 6.0


def make_args(device):
    return argparse.Namespace(
        device=device, mode="learned", pipeline="hybrid", edge_mlp_type="GCN", conditional=True, sparse_edge_mlp=True,
        t_init=0.7, t_min=0.5, degree_bias_coef=0.3, reg1=True, reg2=True, regularizer1_coef=1.0, consist_reg_coef=0.5,
        hybrid_checkpoint=True, drop_rate=0.3, lr=1e-3)


def build_model(S, device, fused=True):
    """Same two Adam optimisers over the same (overlapping) parameter sets as main.py:100,122; `fused` selects
    sgs_gnn_amd.FusedAdam (one launch per group, torch.optim.Adam's update rule and state layout, capturable) so that a
    replayed step includes its optimiser steps; 0 = torch.optim.Adam (foreach), stepped eagerly after each replay."""
    torch.manual_seed(42)
    m = S.GNNModel(NFEAT, HID, NCLS, dropout_prob=0.3, edge_mlp_type="GCN").to(device)
    Adam = S.FusedAdam if fused else torch.optim.Adam
    opt_gnn = Adam([p for n, p in m.named_parameters() if "gcn" in n], lr=1e-3)             # main.py:100
    opt_edge = Adam([p for n, p in m.named_parameters() if "edge_prob_mlp" in n], lr=1e-3)  # main.py:122
    opt_all = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=5e-4)                               # main.py:123
    return m, opt_gnn, opt_edge, opt_all


def kernel_roofline(S, model, batch, reps):
    """Live HIP-event timing of the dominant kernel (the fused MFMA edge scorer, forward over all E
    candidate edges of a partition) on the stream it is launched on (torch's current stream)."""
    ops = S.ops
    E = batch.edge_index.shape[1]
    H = HID
    sc = model.edge_prob_mlp
    codes = torch.relu(torch.randn(N_NODES, H, device=batch.x.device))
    L = S._lib.lib()
    U = (codes @ sc.fc1.weight[:, H:].t()).contiguous()
    out = torch.empty(E, dtype=torch.float32, device=codes.device)
    ws = ops.workspace(L.sgs_edge_score_workspace_bytes(N_NODES, H, E), codes.device)
    W1, b1, w2, b2 = sc.fc1.weight.detach().contiguous(), sc.fc1.bias.detach(), sc.fc2.weight.detach().reshape(-1).contiguous(), sc.fc2.bias.detach()

    def launch():
        S._lib.check(L.sgs_edge_score_fwd(codes.data_ptr(), U.data_ptr(), N_NODES, H, batch.edge_index.data_ptr(), E, 0, W1.data_ptr(),
                                          b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), 0.3, 1, 2, out.data_ptr(), ws.data_ptr(),
                                          ws.numel(), torch.cuda.current_stream().cuda_stream), "edge_score_fwd")
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        launch()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    flops = E * (2.0 * H * H + 2.0 * H)          # algorithmic flops per launch after the W1 split
    achieved = flops / (ms * 1e-3) / 1e12
    # HBM bytes per launch come from the separate rocprofv3 --pmc passes on this very kernel and shape
    # (profiles/r01_scorer_pmc.json: 2 x FETCH_SIZE + WRITE_SIZE, gfx950 correction applied); null if the
    # resident shape differs from the profiled one.
    traffic = None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_scorer_pmc.json")))
        if f"E={E}," in pmc["kernel"]:
            traffic = pmc["hbm_traffic_bytes_per_launch"]
    except Exception:
        pass
    return {"bound": "mfma", "kernel": "edge_score_kernel<8,false,false> (sgs_edge_score_fwd)", "achieved": round(achieved, 3),
            "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / F32_MFMA_PEAK_TFLOPS, 4),
            "traffic": traffic, "edges_per_launch": E, "ms_per_launch": round(ms, 4),
            "flops_per_edge": 2 * H * H + 2 * H}


def cpu_baseline(batch_cpu, steps=2):
    """The oracle's hybrid step (the reference's op sequence on CPU: materialised [E,2H] scorer,
    torch topk draw, gather->mul->index_add GCN, losses, autograd, two Adam steps) on the host
    cores, on a bounded sample of the same workload."""
    from oracle import sgs_oracle as O
    threads = min(os.cpu_count(), 16)          # the GPU box's CPU share for one GPU
    torch.set_num_threads(threads)
    P = {k: v.requires_grad_(True) for k, v in O.init_params(NFEAT, HID, NCLS, "GCN", seed=1).items()}
    cfg = O.StepConfig(pipeline="hybrid", scorer="GCN", q=Q, conditional=True, drop_rate=0.3)
    b = dict(x=batch_cpu.x, edge_index=batch_cpu.edge_index, y=batch_cpu.y, train_mask=batch_cpu.train_mask, prob=batch_cpu.prob)
    E = batch_cpu.edge_index.shape[1]
    g = torch.Generator().manual_seed(0)
    st_e, st_g = {}, {}

    def one():
        nz = O.StepNoise(prior_noise=torch.empty(E).exponential_(1, generator=g), sample_noise=torch.empty(E).exponential_(1, generator=g))
        nz.masks_pass1 = O.Masks(enc_hidden=torch.rand(N_NODES, HID, generator=g) > 0.3, score_hidden=torch.rand(E, HID, generator=g) > 0.3)
        nz.gnn_keep_learned = torch.rand(N_NODES, HID, generator=g) > 0.3
        nz.gnn_keep_random = torch.rand(N_NODES, HID, generator=g) > 0.3
        R = O.learned_step_forward(P, b, cfg, nz)
        for p_ in P.values():
            p_.grad = None
        R["loss"].backward()
        grads = {k: v.grad for k, v in P.items()}
        with torch.no_grad():
            if R["update_edge_mlp"]:
                O.adam_step({k: v for k, v in P.items() if "edge_prob_mlp" in k}, grads, st_e)
            O.adam_step({k: v for k, v in P.items() if "gcn" in k}, grads, st_g)
    one()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    dt = time.perf_counter() - t0
    return {"value": round(Q * steps / dt, 1), "unit": "sampled edges/s", "steps_per_s": round(steps / dt, 4), "cores": threads,
            "kind": "port", "sample": f"{steps} hybrid steps (after 1 warm-up) on one synthetic Reddit-like partition, "
                                      f"n={N_NODES}, E={E}, q={Q}, F={NFEAT}, H={HID}, fp32, torch {torch.__version__} CPU"}


def pool_indices(sizes, rank, world, per_rank):
    """Which partitions of the common stream rank `rank` holds, in step order.  Data-parallel steps end in a gradient all-reduce, so
    a step lasts as long as its slowest rank: the stream is dealt out BY SIZE (sorted by edge count, rank r takes every world-th
    one), so that the partitions the ranks process in the same step have adjacent sizes -- in particular all sampled or all
    unsampled -- and the step order is then shuffled with a permutation common to all ranks."""
    import random
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--pool", type=int, default=12, help="distinct partition batches kept resident per rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fused-adam", type=int, default=1, help="1 (default): sgs_gnn_amd.FusedAdam (one launch per group, captured with the step); 0: torch.optim.Adam (foreach, eager)")
    ap.add_argument("--score-variant", type=int, default=-1, help="scorer forward kernel for the timed steps (benchmarking A/B; -1 = library default)")
    ap.add_argument("--hipgraph", type=int, default=1, help="1 (default): replay each partition's step from captured HIP graphs "
                    "(stepgraph.py; every pool partition is visited twice -- eager, capture -- before the W warm-up steps); 0: eager launches")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
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
    model, opt_gnn, opt_edge, opt_all = build_model(S, device, fused=bool(a.fused_adam))
    crit = torch.nn.CrossEntropyLoss()
    args = make_args(device)

    # partition pool (per rank: its own shard of the stream), resident in HBM before timing
    pool = make_pool(S, rank, world, a.pool, device)
    warm = [pool[i % len(pool)] for i in range(a.warmup)]
    if a.hipgraph:
        args.sgs_hipgraph = True
        args.sgs_dp_global_gate = world > 1      # N > 1: one gate per step over the union of the ranks' batches (dist.py)
        warm = list(pool) * 2 + warm
    timed = [pool[i % len(pool)] for i in range(a.steps)]
    sampled = sum(Q for b in timed if b.edge_index.shape[1] > Q)

    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        if warm:
            S.train(args, 0, 10, model, opt_gnn, opt_edge, opt_all, crit, warm, q=Q, alternate_frequency=0)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ret = S.train(args, 1, 10, model, opt_gnn, opt_edge, opt_all, crit, timed, q=Q, alternate_frequency=0)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0

    t = torch.tensor([dt, float(sampled)], dtype=torch.float64, device=device)
    if world > 1:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt_all, sampled_all = float(tmax[0]), float(tsum[1])
    else:
        dt_all, sampled_all = dt, float(sampled)

    if rank == 0:
        big = max(pool, key=lambda b: b.edge_index.shape[1])
        L = S._lib.lib()
        alts = {}
        names = {0: "lds_tiled", 1: "stream_32_edge_wave_tile", 2: "weight_stationary_persistent", 3: "stream_64_edge_wave_tile",
                 4: "bf16x6_split_on_bf16_mfma"}
        used = a.score_variant if a.score_variant >= 0 else 4           # automatic choice at this E (>= 65 536 edges) and H = 256
        for v, name in names.items():                                      # in-process A/B of the scorer forward kernels
            if v == used:
                continue
            L.sgs_edge_score_set_variant(v)
            r_ = kernel_roofline(S, model, big, reps=20)
            alts[name] = {"achieved": r_["achieved"], "ms_per_launch": r_["ms_per_launch"]}
        L.sgs_edge_score_set_variant(a.score_variant)
        roof = kernel_roofline(S, model, big, reps=20)       # the variant used by the timed steps above
        roof["kernel"] = f"sgs_edge_score_fwd, forward variant {used} ({names[used]})"
        if used == 4:
            roof["peak"] = round(BF16X6_PEAK_TFLOPS, 1)
            roof["frac"] = round(roof["achieved"] / BF16X6_PEAK_TFLOPS, 4)
            roof["peak_note"] = ("algorithmic fp32 flops; the kernel runs 6 bf16 MFMA products per fp32 product (exact 3-way splits, "
                                 "fp32-faithful), so peak = dense bf16 MFMA peak 2500 / 6")
            roof["bf16_mfma_tflops_executed"] = round(6 * roof["achieved"], 1)
            roof["vs_fp32_mfma_peak"] = round(roof["achieved"] / F32_MFMA_PEAK_TFLOPS, 4)
        roof["alt_variants"] = alts
        rec = {
            "metric": "sampled edges/sec + training steps/sec, Reddit hybrid 20% sparsity",
            "value": round(sampled_all / dt_all, 1), "unit": "sampled edges/s",
            "steps_per_s": round(a.steps * world / dt_all, 3),
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt_all / a.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "dtype_note": "fp32 tensors, accumulators and results throughout; the scorer's fp32 contractions (forward, and the backward's "
                          "recompute / dv.W1a / tall weight-gradient GEMM) run as six bf16 MFMA products over exact 3-way operand splits "
                          "(error measured at the fp32-MFMA kernels' level, see roofline.peak_note)",
            "data": "synthetic",
            "config": {"workload": "Reddit-like METIS partition stream (S3): n=1013 F=602 C=41 H=256, E_b in [60k,500k] "
                                   "(52% above q), q=100000, hybrid pipeline, EdgeProbGCN scorer, conditional gate, reg1+reg2, "
                                   "dropout 0.3, Adam x2", "pool": a.pool, "partitions_above_q": sum(1 for b in timed if b.edge_index.shape[1] > Q),
                       "parallelism": f"dp{world} (partition-sharded by size across ranks, global gate, 1 flat gradient all-reduce/step)" if world > 1 else "single",
                       "hipgraph_replay": bool(getattr(args, "sgs_hipgraph", False)),
                       "adam": "sgs_gnn_amd.FusedAdam (in-graph)" if a.fused_adam else "torch.optim.Adam (foreach, eager)"},
            "mean_loss": round(ret[0], 5), "conditional_updates": ret[2], "total_updates": ret[3],
            "roofline": roof,
        }
        if not a.no_cpu_baseline and world == 1:
            above = [b for b in pool if b.edge_index.shape[1] > Q]
            cpu_b = min(above, key=lambda b: b.edge_index.shape[1]) if above else big      # smallest sampled partition: bounded CPU time
            rec["cpu_baseline"] = cpu_baseline(cpu_b.to("cpu"))
        print(json.dumps(rec))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
