"""GPU + gloo (R processes sharing cuda:0): the edge-partitioned path of config 5.
  * one exact global top-q over edge-sharded keys selects the SAME edge set for 1, 2 and 3 ranks and
    equals the single-GPU fused sampler bit for bit (noise keyed by global edge id, normaliser reduced
    in the single-GPU order, integer histogram all-reduce);
  * the edge-sharded evaluate forward (partial aggregates + all-reduce of node embeddings) reproduces
    the single-rank logits."""
import argparse
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_inputs():
    import sgs_gnn_amd as S
    b = S.synthetic_graph(300, 21000, 16, 5, seed=3, train_frac=0.5, device=DEV)
    E = b.edge_index.shape[1]
    g = torch.Generator(device=DEV).manual_seed(9)
    p = torch.sigmoid(torch.randn(E, device=DEV, generator=g))
    noise = torch.empty(E, device=DEV).exponential_(1, generator=g)
    return S, b, p, noise


def _worker(rank, world, port, q_out):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from importlib import import_module
        S, b, p, noise = _make_inputs()
        sh_mod = import_module("sgs_gnn_amd.sharded")
        shard = sh_mod.EdgeShard(b, rank, world)
        lo, hi = shard.bounds[rank], shard.bounds[rank + 1]
        E = b.edge_index.shape[1]
        q = E // 5
        res = {}
        for name, mode, prior, nz in [("learned_noise", S.ops.SAMPLE_LEARNED, shard.prob, noise[lo:hi].contiguous()),
                                      ("learned_seed", S.ops.SAMPLE_LEARNED, shard.prob, None),
                                      ("istest_seed", S.ops.SAMPLE_LEARNED, None, None),
                                      ("prior_seed", S.ops.SAMPLE_PRIOR, None, None)]:
            src = shard.prob if mode == S.ops.SAMPLE_PRIOR else p[lo:hi].contiguous()
            r = sh_mod.dist_sample_topq(mode, src, prior, 0.3, q, shard.edge_index, shard.edge_offset, shard.bounds, noise_local=nz,
                                        seed=77, stream_id=5)
            res[name] = dict(mask=r.mask.cpu().numpy(), eid=r.eid.cpu().numpy(), sei=r.edge_index.cpu().numpy(), stats=r.stats.cpu().numpy())
        torch.manual_seed(0)
        m = S.GNNModel(16, 64, 5, dropout_prob=0.3, edge_mlp_type="GCN").to(DEV).eval()
        args = argparse.Namespace(degree_bias_coef=0.3)
        out, smp = sh_mod.sharded_evaluate_forward(args, m, shard, q, seed=123, stream_id=2)
        res["fwd"] = dict(out=out.cpu().numpy(), mask=smp.mask.cpu().numpy())
        q_out.put((rank, res))
    finally:
        dist.destroy_process_group()


def _run(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return [{k: {kk: torch.from_numpy(vv) for kk, vv in v.items()} for k, v in got[r].items()} for r in range(world)]


def test_sharded_draw_and_forward_are_rank_count_invariant():
    S, b, p, noise = _make_inputs()
    E = b.edge_index.shape[1]
    q = E // 5
    # single-GPU fused sampler = the reference point
    ref = {
        "learned_noise": S.ops.sample_topq(S.ops.SAMPLE_LEARNED, p, b.prob, 0.3, q, b.edge_index, noise=noise),
        "learned_seed": S.ops.sample_topq(S.ops.SAMPLE_LEARNED, p, b.prob, 0.3, q, b.edge_index, seed=77, stream_id=5),
        "istest_seed": S.ops.sample_topq(S.ops.SAMPLE_LEARNED, p, None, 0.3, q, b.edge_index, seed=77, stream_id=5),
        "prior_seed": S.ops.sample_topq(S.ops.SAMPLE_PRIOR, b.prob, None, 0.0, q, b.edge_index, seed=77, stream_id=5),
    }
    runs = {w: _run(w) for w in (1, 2, 3)}
    for name, r in ref.items():
        for w, parts in runs.items():
            mask = torch.cat([pt[name]["mask"] for pt in parts])
            eid = torch.cat([pt[name]["eid"] for pt in parts])
            sei = torch.cat([pt[name]["sei"] for pt in parts], dim=1)
            assert int(mask.sum()) == q, (name, w)
            assert torch.equal(mask, r.mask.cpu()), (name, w)
            assert torch.equal(eid, r.eid.cpu()) and torch.equal(sei, r.edge_index.cpu()), (name, w)
            for pt in parts:                                   # identical stats on every rank, Z bit-equal to the fused path
                assert torch.equal(pt[name]["stats"][:3], r.stats.cpu()[:3]), (name, w)
    base = runs[1][0]["fwd"]
    for w in (2, 3):
        parts = runs[w]
        assert torch.equal(torch.cat([pt["fwd"]["mask"] for pt in parts]), base["mask"])
        for pt in parts:
            torch.testing.assert_close(pt["fwd"]["out"], base["out"], rtol=1e-4, atol=1e-5)


def _dp_worker(rank, world, port, q_out, hipgraph=False, global_gate=False, uneven=False, empty=False):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import contextlib
        import io
        import sgs_gnn_amd as S
        S.fix_seeds(100 + rank)                                   # different noise / dropout streams per rank
        torch.manual_seed(0)                                      # identical initial replicas
        m = S.GNNModel(12, 32, 5, dropout_prob=0.3, edge_mlp_type="GCN").to(DEV)
        Adam = S.FusedAdam if hipgraph else torch.optim.Adam      # graph mode as bench.py runs it: FusedAdam, stepped eagerly after the all-reduce
        opt_gnn = Adam([p for n, p in m.named_parameters() if "gcn" in n], lr=1e-2)
        opt_edge = Adam([p for n, p in m.named_parameters() if "edge_prob_mlp" in n], lr=1e-2)
        opt_all = torch.optim.Adam(m.parameters(), lr=1e-2)
        # rank-specific partitions, one of them too small to be sampled (E <= q): exercises every sync branch
        sizes = [6000, 900, 5000] if rank == 0 else [5500, 7000, 800]
        if uneven and rank == 0:
            sizes = sizes + [6500]                                # 4 batches against 3: P % world != 0 (dist.shard_batches)
        batches = [S.synthetic_graph(300, e, 12, 5, seed=10 * rank + i, train_frac=0.5, device=DEV) for i, e in enumerate(sizes)]
        if uneven and rank == 1:
            batches[1].train_mask = torch.zeros_like(batches[1].train_mask)     # a batch the trainer skips on this rank only
        if empty and rank == 1:
            # a partition without a single intra-partition edge: the captured slots cannot take it, the step is launched eagerly
            batches[0].edge_index = torch.zeros(2, 0, dtype=torch.int64, device=DEV)
            batches[0].prob = torch.zeros(0, dtype=torch.float32, device=DEV)
        args = argparse.Namespace(device=DEV, mode="learned", pipeline="hybrid", conditional=True, sparse_edge_mlp=True, t_init=0.7,
                                  t_min=0.5, degree_bias_coef=0.3, reg1=True, reg2=True, regularizer1_coef=1.0,
                                  consist_reg_coef=0.5, hybrid_checkpoint=False, sgs_hipgraph=hipgraph, sgs_dp_global_gate=global_gate)
        with contextlib.redirect_stdout(io.StringIO()):
            for ep in range(4):
                ret = S.train(args, ep, 4, m, opt_gnn, opt_edge, opt_all, torch.nn.CrossEntropyLoss(), batches, q=1000)
        q_out.put((rank, {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}, ret[2], ret[3]))
    finally:
        dist.destroy_process_group()


def test_data_parallel_global_gate_keeps_replicas_identical():
    """args.sgs_dp_global_gate (what bench.py --gpus N sets): one gate per step over the union of the ranks' batches -- the four
    counts are summed over ranks, every rank takes the same branch; ranks whose partition is not sampled join the collective
    with zeros (the test's ranks mix sampled and unsampled partitions in the same step).  Replicas stay bit-identical."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q, True, True)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        rank, sd, cond, tot = q.get(timeout=120)
        got[rank] = (sd, cond, tot)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert got[0][2] == got[1][2] == 3
    for k in got[0][0]:
        a, b = torch.from_numpy(got[0][0][k]), torch.from_numpy(got[1][0][k])
        assert torch.equal(a, b), k
        assert bool(torch.isfinite(a).all())


@pytest.mark.parametrize("hipgraph", [False, True])
def test_data_parallel_uneven_shards_finish_with_null_steps(hipgraph):
    """Ranks with different step counts (4 usable batches on rank 0; 3 on rank 1, one of them without train nodes -> 2 steps): the
    epoch agrees on the longest shard and the short rank joins the remaining steps' collectives with zero gradients
    (training._null_step) -- no hang, replicas bit-identical."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q, hipgraph, False, True)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        rank, sd, cond, tot = q.get(timeout=300)
        got[rank] = (sd, cond, tot)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert got[0][2] == 4 and got[1][2] == 2
    for k in got[0][0]:
        a, b = torch.from_numpy(got[0][0][k]), torch.from_numpy(got[1][0][k])
        assert torch.equal(a, b), k
        assert bool(torch.isfinite(a).all())


@pytest.mark.parametrize("global_gate", [False, True])
def test_data_parallel_graph_mode_with_a_batch_the_slots_cannot_take(global_gate):
    """Graph-mode data parallel where one rank meets a partition without edges (stepgraph._EagerHandle): that rank's eager step must
    issue exactly the collectives its peer's replayed step issues ([gate sum,] ONE bucket all-reduce, then the shared optimiser
    graph) -- no hang, no mixed-up reductions, replicas bit-identical."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q, True, global_gate, False, True)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        rank, sd, cond, tot = q.get(timeout=300)
        got[rank] = (sd, cond, tot)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert got[0][2] == got[1][2] == 3
    for k in got[0][0]:
        a, b = torch.from_numpy(got[0][0][k]), torch.from_numpy(got[1][0][k])
        assert torch.equal(a, b), k
        assert bool(torch.isfinite(a).all())


@pytest.mark.parametrize("hipgraph", [False, True])
def test_data_parallel_train_keeps_replicas_identical(hipgraph):
    """N > 1 on the partition stream: per-rank batches, one flat gradient all-reduce per step, the scorer's
    optimiser steps on every rank iff any rank's gate chose 'learned' -> replicas remain bit-identical.
    hipgraph=True: the same with each rank replaying its partitions' steps from captured HIP graphs (every step a replay, from
    the first one on); the collectives stay between the replays."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q, hipgraph)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        rank, sd, cond, tot = q.get(timeout=300)
        got[rank] = (sd, cond, tot)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert got[0][2] == got[1][2] == 3
    for k in got[0][0]:
        a, b = torch.from_numpy(got[0][0][k]), torch.from_numpy(got[1][0][k])
        assert torch.equal(a, b), k
        assert bool(torch.isfinite(a).all())


def _train_setup():
    import sgs_gnn_amd as S
    b = S.synthetic_graph(300, 21000, 16, 5, seed=3, train_frac=0.5, device=DEV)
    torch.manual_seed(0)
    m = S.GNNModel(16, 64, 5, dropout_prob=0.3, edge_mlp_type="GCN").to(DEV)
    og = torch.optim.Adam([p for n, p in m.named_parameters() if "gcn" in n], lr=1e-3)
    oe = torch.optim.Adam([p for n, p in m.named_parameters() if "edge_prob_mlp" in n], lr=1e-3)
    args = argparse.Namespace(device=DEV, mode="learned", pipeline="hybrid", conditional=True, sparse_edge_mlp=True, t_init=0.7,
                              t_min=0.5, degree_bias_coef=0.3, reg1=True, reg2=True, regularizer1_coef=1.0, consist_reg_coef=0.5,
                              hybrid_checkpoint=False)
    return S, b, m, og, oe, args


def _sharded_train_worker(rank, world, port, q_out, blocks=False):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from importlib import import_module
        S, b, m, og, oe, args = _train_setup()
        sh = import_module("sgs_gnn_amd.sharded")
        shard = sh.EdgeShard(b, rank, world)
        q = b.edge_index.shape[1] // 5
        S.fix_seeds(5)
        out = {}
        for step in range(2):
            step_fn = sh.train_step_blocksharded if blocks else sh.train_step_sharded
            tr = step_fn(args, m, shard, og, oe, torch.nn.CrossEntropyLoss(), q)
            if step == 0:
                out["grads"] = {k: (v.grad.detach().cpu().numpy() if v.grad is not None else None) for k, v in m.named_parameters()}
                out["mask"] = tr["sample"].mask.cpu().numpy()
                out["rmask"] = tr["random"].mask.cpu().numpy()
                out["logits"] = tr["learned_out"].cpu().numpy()
                out["loss0"] = float(tr["loss"])
                out["upd0"] = bool(tr["update_edge_mlp"])
        out["params"] = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
        q_out.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("blocks", [False, True])
def test_sharded_training_step_matches_single_gpu_train(blocks):
    """Config 5 training: f/g operators at the shard boundaries give every rank the complete gradients; the
    2- and 3-rank steps reproduce the regular single-GPU train() (same seeds: noise and dropout are keyed
    by global ids) and the replicas are bit-identical to each other.  blocks=True: the node-block form (reduce-scatter forward /
    all-gather backward, node-level work on 1 / R of the rows, one flat parameter-gradient all-reduce) against the same reference."""
    import contextlib
    import io
    S, b, m, og, oe, args = _train_setup()
    q = b.edge_index.shape[1] // 5
    S.fix_seeds(5)
    oa = torch.optim.Adam(m.parameters(), lr=1e-3)
    args._sgs_trace = tr = {}
    ref = {}
    with contextlib.redirect_stdout(io.StringIO()):
        for step in range(2):
            S.train(args, step, 2, m, og, oe, oa, torch.nn.CrossEntropyLoss(), [b], q=q)
            if step == 0:
                ref["grads"] = {k: (v.grad.detach().cpu() if v.grad is not None else None) for k, v in m.named_parameters()}
                ref["mask"], ref["logits"] = tr["sample"].mask.cpu(), tr["learned_out"].cpu()
                ref["loss0"], ref["upd0"] = float(tr["loss"]), bool(tr["update_edge_mlp"])
    ref["params"] = {k: v.detach().cpu() for k, v in m.state_dict().items()}

    for world in (2, 3):
        ctx = mp.get_context("spawn")
        qq = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_sharded_train_worker, args=(r, world, port, qq, blocks)) for r in range(world)]
        for p in procs:
            p.start()
        got = dict(qq.get(timeout=300) for _ in range(world))
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
        mask = torch.cat([torch.from_numpy(got[r]["mask"]) for r in range(world)])
        assert torch.equal(mask, ref["mask"]), world
        for r in range(world):
            o = got[r]
            assert o["upd0"] == ref["upd0"]
            assert abs(o["loss0"] - ref["loss0"]) < 2e-5
            torch.testing.assert_close(torch.from_numpy(o["logits"]), ref["logits"], rtol=1e-4, atol=1e-5)
            for k, g in ref["grads"].items():
                if g is None:
                    assert o["grads"][k] is None or float(abs(o["grads"][k]).max()) == 0.0, k
                else:
                    a = torch.from_numpy(o["grads"][k])
                    err = float((a - g).abs().max()) / (float(g.abs().max()) + 1e-12)
                    assert err < 2e-3, (world, k, err)
            for k, v in ref["params"].items():                       # two Adam steps: lr-sized moves, compare loosely
                torch.testing.assert_close(torch.from_numpy(o["params"][k]), v, rtol=0, atol=2.5e-3)
            for k in o["params"]:                                     # replicas are bit-identical to each other
                assert (o["params"][k] == got[0]["params"][k]).all(), (world, k)


def _s5_setup(path=None):
    """Config 5's graph: the full Reddit node count and candidate-edge count (N = 232 965, F = 602, C = 41, E = 114.6 M both
    directions stored, q = 20 %), synthetic look-alike (power-law degrees), dropout 0.3, conditional gate, both regularisers.
    `path`: load the graph another process saved (torch's GPU generators do not reproduce a graph of this size bit for bit
    across processes, so the ranks must not each build their own)."""
    import sgs_gnn_amd as S
    if path is None:
        b = S.synthetic_graph(232_965, 114_615_892, 602, 41, seed=77, train_frac=0.66, power=0.35, device=DEV)     # max degree ~2e4, as Reddit (21 657)
    else:
        b = S.Batch(**{k: v.to(DEV) for k, v in torch.load(path, weights_only=True).items()})
    torch.manual_seed(0)
    m = S.GNNModel(602, 256, 41, dropout_prob=0.3, edge_mlp_type="GCN").to(DEV)
    og = torch.optim.Adam([p for n, p in m.named_parameters() if "gcn" in n], lr=1e-3)
    oe = torch.optim.Adam([p for n, p in m.named_parameters() if "edge_prob_mlp" in n], lr=1e-3)
    args = argparse.Namespace(device=DEV, mode="learned", pipeline="hybrid", conditional=True, sparse_edge_mlp=True, t_init=0.7,
                              t_min=0.5, degree_bias_coef=0.3, reg1=True, reg2=True, regularizer1_coef=1.0, consist_reg_coef=0.5,
                              hybrid_checkpoint=True)
    return S, b, m, og, oe, args


def _s5_worker(rank, world, port, q_out, path, blocks=False):
    import sys
    import numpy as np
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from importlib import import_module
        S, b, m, og, oe, args = _s5_setup(path)
        sh = import_module("sgs_gnn_amd.sharded")
        shard = sh.EdgeShard(b, rank, world)
        q = b.edge_index.shape[1] // 5
        del b
        S.fix_seeds(5)
        tr = (sh.train_step_blocksharded if blocks else sh.train_step_sharded)(args, m, shard, og, oe, torch.nn.CrossEntropyLoss(), q)
        out = dict(mask=np.packbits(tr["sample"].mask.cpu().numpy()), rmask=np.packbits(tr["random"].mask.cpu().numpy()),
                   n_local=int(shard.edge_index.shape[1]), loss=float(tr["loss"]), upd=bool(tr["update_edge_mlp"]))
        if rank == 0:
            out["logits"] = tr["learned_out"].cpu().numpy()
            out["grads"] = {k: (v.grad.detach().cpu().numpy() if v.grad is not None else None) for k, v in m.named_parameters()}
        q_out.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("blocks", [False, True])
def test_s5_full_reddit_scale_edge_sharded_step_matches_single_gpu(blocks):
    """(blocks=True: the node-block form of the step -- reduce-scatter forward / all-gather backward -- to the same bounds.)
    BASELINE.json config 5 at its size: one hybrid training step on the full-Reddit-sized graph, edge-partitioned over two
    ranks (gloo, both on this one MI355X; the embedding all-reduces move 238 MB each), against the single-GPU train() on the
    same graph: both draws select the same 22.9 M edges bit for bit (global exponential race over the shards), same gate,
    logits within 1e-4, gradients within 2e-3 of each tensor's largest entry.
    The prior draw depends on `prob` and the noise alone and must be bit-identical.  The learned draw races keys built from the
    scorer's probabilities, which pass through aggregations whose fp32 summation order differs between one rank (row sums) and
    two (partial sums, then the all-reduce); its sets are therefore compared by their symmetric difference, bounded at 2e-5 of
    q -- measured: 0 of 22 923 178.  (The ranks load ONE saved graph: torch's GPU generators do not rebuild a graph of this
    size bit for bit in another process.)"""
    import contextlib
    import io
    import numpy as np
    S, b, m, og, oe, args = _s5_setup()
    E = b.edge_index.shape[1]
    q = E // 5
    path = f"/dev/shm/sgs_s5_{os.getpid()}.pt"
    torch.save({k: getattr(b, k).cpu() for k in ("x", "edge_index", "y", "train_mask", "val_mask", "test_mask", "prob")}, path)
    try:
        _s5_compare(S, b, m, og, oe, args, E, q, path, blocks)
    finally:
        os.remove(path)


def _s5_compare(S, b, m, og, oe, args, E, q, path, blocks=False):
    import contextlib
    import io
    import numpy as np
    S.fix_seeds(5)
    args._sgs_trace = tr = {}
    with contextlib.redirect_stdout(io.StringIO()):
        S.train(args, 0, 2, m, og, oe, None, torch.nn.CrossEntropyLoss(), [b], q=q)
    ref = dict(mask=np.packbits(tr["sample"].mask.cpu().numpy()), rmask=np.packbits(tr["prior_sample"].mask.cpu().numpy()),
               logits=tr["learned_out"].cpu(), loss=float(tr["loss"]),
               upd=bool(tr["update_edge_mlp"]), grads={k: (v.grad.detach().cpu() if v.grad is not None else None) for k, v in m.named_parameters()})
    del tr, m, og, oe, b
    args._sgs_trace = None
    torch.cuda.empty_cache()

    world = 2
    ctx = mp.get_context("spawn")
    qq = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_s5_worker, args=(r, world, port, qq, path, blocks)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(qq.get(timeout=900) for _ in range(world))
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    # shards are 2048-aligned, so the packed per-rank masks concatenate on byte boundaries
    assert got[0]["n_local"] % 8 == 0 and got[0]["n_local"] + got[1]["n_local"] == E
    assert np.array_equal(np.concatenate([got[0]["rmask"], got[1]["rmask"]]), ref["rmask"])          # prior draw: bit-identical
    mask = np.concatenate([got[0]["mask"], got[1]["mask"]])
    flips = int(np.unpackbits(mask ^ ref["mask"]).sum())
    print("learned-draw symmetric difference:", flips, "of", q)
    assert int(np.unpackbits(mask)[:E].sum()) == q and flips <= 2e-5 * q, flips          # (measured: see the test's output)
    for r in range(world):
        assert got[r]["upd"] == ref["upd"]
        assert abs(got[r]["loss"] - ref["loss"]) < 1e-4
    torch.testing.assert_close(torch.from_numpy(got[0]["logits"]), ref["logits"], rtol=1e-4, atol=1e-4)
    for k, g in ref["grads"].items():
        a = got[0]["grads"][k]
        if g is None:
            assert a is None or float(abs(a).max()) == 0.0, k
        else:
            err = float((torch.from_numpy(a) - g).abs().max()) / (float(g.abs().max()) + 1e-12)
            assert err < 2e-3, (k, err)


def _rccl_worker(port, q_out):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", SGS_DP_FORCE="1",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        import contextlib
        import io
        from importlib import import_module
        import sgs_gnn_amd as S
        D = import_module("sgs_gnn_amd.dist")
        assert D.is_parallel() and dist.get_backend() == "nccl"
        out = {}
        # (i) graph-mode data parallel: bucket all-reduce between replays, gate sum, shared optimiser graph -- through RCCL
        S.fix_seeds(100)
        torch.manual_seed(0)
        m = S.GNNModel(12, 32, 5, dropout_prob=0.3, edge_mlp_type="GCN").to(DEV)
        og = S.FusedAdam([p for n, p in m.named_parameters() if "gcn" in n], lr=1e-2)
        oe = S.FusedAdam([p for n, p in m.named_parameters() if "edge_prob_mlp" in n], lr=1e-2)
        batches = [S.synthetic_graph(300, e, 12, 5, seed=i, train_frac=0.5, device=DEV) for i, e in enumerate([6000, 900, 5000, 7000])]
        args = argparse.Namespace(device=DEV, mode="learned", pipeline="hybrid", conditional=True, sparse_edge_mlp=True, t_init=0.7,
                                  t_min=0.5, degree_bias_coef=0.3, reg1=True, reg2=True, regularizer1_coef=1.0,
                                  consist_reg_coef=0.5, hybrid_checkpoint=False, sgs_hipgraph=True, sgs_dp_global_gate=True)
        with contextlib.redirect_stdout(io.StringIO()):
            for ep in range(4):
                ret = S.train(args, ep, 4, m, og, oe, None, torch.nn.CrossEntropyLoss(), batches, q=1000)
        sg = m._sgs_stepgraphs
        out["dp"] = dict(ret=ret, dp=bool(sg.dp), g3=sg.g3 is not None,
                         finite=all(bool(torch.isfinite(p).all()) for p in m.parameters()),
                         state={k: v.detach().cpu().numpy() for k, v in m.state_dict().items()})
        # (ii) the edge-sharded training step (config 5) with every collective issued (world size 1)
        sh = import_module("sgs_gnn_amd.sharded")
        S2, b, m2, og2, oe2, args2 = _train_setup()
        shard = sh.EdgeShard(b, 0, 1)
        S.fix_seeds(5)
        tr = sh.train_step_sharded(args2, m2, shard, og2, oe2, torch.nn.CrossEntropyLoss(), b.edge_index.shape[1] // 5)
        out["sharded"] = dict(loss=float(tr["loss"]), mask=tr["sample"].mask.cpu().numpy(), logits=tr["learned_out"].cpu().numpy())
        # (iii) the node-block form: RCCL reduce_scatter / all_gather of padded blocks, halo exchange, flat gradient all-reduce
        S3, b3, m3, og3, oe3, args3 = _train_setup()
        S.fix_seeds(5)
        tr = sh.train_step_blocksharded(args3, m3, sh.EdgeShard(b3, 0, 1), og3, oe3, torch.nn.CrossEntropyLoss(), b3.edge_index.shape[1] // 5)
        out["blocks"] = dict(loss=float(tr["loss"]), mask=tr["sample"].mask.cpu().numpy(), logits=tr["learned_out"].cpu().numpy())
        q_out.put(out)
    finally:
        dist.destroy_process_group()


def test_rccl_backend_runs_the_data_parallel_and_the_sharded_step():
    """backend "nccl" (= RCCL on ROCm), world size 1, SGS_DP_FORCE=1: RCCL communicator initialisation on this box, device-tensor
    all-reduces on RCCL's stream between HIP-graph replays (captures in thread-local mode beside the watchdog thread), the gate
    sum, the shared optimiser graph and every collective of the edge-sharded step.  With one rank every collective is the
    identity, so the results must equal the plain single-process runs."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    out = q.get(timeout=300)
    p.join(120)
    assert p.exitcode == 0
    assert out["dp"]["dp"] and out["dp"]["g3"] and out["dp"]["finite"] and out["dp"]["ret"][3] == 4
    # the sharded step against the regular single-GPU train() (same seeds: noise and dropout are keyed by global edge id)
    S, b, m, og, oe, args = _train_setup()
    S.fix_seeds(5)
    args._sgs_trace = tr = {}
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        S.train(args, 0, 10, m, og, oe, None, torch.nn.CrossEntropyLoss(), [b], q=b.edge_index.shape[1] // 5)
    assert torch.equal(torch.from_numpy(out["sharded"]["mask"]), tr["sample"].mask.cpu())
    torch.testing.assert_close(torch.from_numpy(out["sharded"]["logits"]), tr["learned_out"].cpu(), rtol=1e-4, atol=1e-5)
    assert torch.equal(torch.from_numpy(out["blocks"]["mask"]), tr["sample"].mask.cpu())
    torch.testing.assert_close(torch.from_numpy(out["blocks"]["logits"]), tr["learned_out"].cpu(), rtol=1e-4, atol=1e-5)
    assert abs(out["blocks"]["loss"] - out["sharded"]["loss"]) < 2e-5
