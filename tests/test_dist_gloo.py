"""CPU, world_size 2 over gloo: the N > 1 host logic (partition sharding + the flat-bucket gradient
average with the gate flag).  The HIP compute is not involved here."""
import os

import pytest
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from importlib import import_module
        import sgs_gnn_amd  # noqa: F401
        D = import_module("sgs_gnn_amd.dist")
        assert D.is_parallel()
        assert D.shard_batches(list(range(7)), rank, world) == list(range(7))[rank::world]
        torch.manual_seed(0)
        lin = torch.nn.Linear(5, 3)
        extra = torch.nn.Parameter(torch.zeros(4))          # a parameter with no grad on rank 1
        params = list(lin.parameters()) + [extra]
        sync = D.GradSync(params)
        lin.weight.grad = torch.full((3, 5), float(rank + 1))
        lin.bias.grad = torch.full((3,), 10.0 * (rank + 1))
        if rank == 0:
            extra.grad = torch.ones(4)
        any_learned = int(sync.any_learned(torch.tensor([1 if rank == 0 else 0], dtype=torch.int32)).item()) > 0
        sync.sync()
        ok = (any_learned is True
              and torch.allclose(lin.weight.grad, torch.full((3, 5), 1.5))
              and torch.allclose(lin.bias.grad, torch.full((3,), 15.0))
              and torch.allclose(extra.grad, torch.full((4,), 0.5))
              and lin.weight.grad.data_ptr() == sync.views[0].data_ptr())
        # second step: grads already live in the bucket views for some params, fresh tensors for others
        lin.weight.grad = None
        lin.bias.grad = torch.full((3,), float(rank))
        sync.sync()
        ok = ok and torch.allclose(lin.weight.grad, torch.zeros(3, 5)) and torch.allclose(lin.bias.grad, torch.full((3,), 0.5))
        none_learned = int(sync.any_learned(torch.zeros(1, dtype=torch.int32)).item()) > 0
        q.put((rank, bool(ok), bool(none_learned)))
    finally:
        dist.destroy_process_group()


def test_gradsync_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res == [(0, True, False), (1, True, False)]


def test_shard_bounds_are_chunk_aligned_and_cover_everything():
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import sgs_gnn_amd  # noqa: F401
    from importlib import import_module
    sh = import_module("sgs_gnn_amd.sharded")
    for E in (1, 2047, 2048, 2049, 50000, 114_615_892):
        for world in (1, 2, 3, 8):
            b = sh.shard_bounds(E, world, 2048)
            assert b[0] == 0 and b[-1] == E and len(b) == world + 1
            assert all(b[i] <= b[i + 1] for i in range(world))
            assert all(x % 2048 == 0 or x == E for x in b[:-1])   # interior boundaries on reduction-chunk boundaries (or empty tail shards)


def test_lookahead_iteration_and_workspace_slots():
    """Host logic of the prefix prefetch (stepgraph.py): the epoch loop reads the loader one batch ahead, and work recorded for a
    second stream takes its scratch from its own arena."""
    from sgs_gnn_amd import ops
    from sgs_gnn_amd.training import _with_lookahead
    assert list(_with_lookahead(iter([1, 2, 3]), True)) == [(1, 2), (2, 3), (3, None)]
    assert list(_with_lookahead(iter([7]), True)) == [(7, None)]
    assert list(_with_lookahead(iter([]), True)) == []
    assert list(_with_lookahead([1, 2], False)) == [(1, None), (2, None)]
    # a generator-backed loader is consumed exactly once, in order
    seen = []

    def gen():
        for i in range(4):
            seen.append(i)
            yield i
    pairs = list(_with_lookahead(gen(), True))
    assert [p[0] for p in pairs] == [0, 1, 2, 3] and seen == [0, 1, 2, 3]
    assert ops._ws_slot == 0
    with ops.workspace_slot(1):
        assert ops._ws_slot == 1
        with ops.workspace_slot(2):
            assert ops._ws_slot == 2
        assert ops._ws_slot == 1
    assert ops._ws_slot == 0


def test_bench_pool_is_dealt_out_by_size_across_ranks():
    """bench.py --gpus N: the common partition stream is dealt out by edge count, so the partitions processed in the same
    data-parallel step have adjacent sizes (a step lasts as long as its slowest rank); every partition is used exactly once."""
    import bench as B
    import sgs_gnn_amd as S
    for world in (2, 4, 8):
        sizes = S.reddit_partition_sizes(12 * world, seed=1000, q=B.Q)
        idx = [B.pool_indices(sizes, r, world, 12) for r in range(world)]
        assert sorted(i for l in idx for i in l) == list(range(12 * world))
        mixed = sum(1 for k in range(12) if len({sizes[idx[r][k]] > B.Q for r in range(world)}) > 1)
        assert mixed <= 1
        for k in range(12):
            col = [sizes[idx[r][k]] for r in range(world)]
            assert max(col) <= 1.7 * min(col)
        above = [sum(sizes[i] > B.Q for i in l) for l in idx]
        assert max(above) - min(above) <= 1
    # one rank: the stream itself (what the single-GPU bench line has always used)
    assert S.reddit_partition_sizes(12, seed=1000, q=B.Q) == S.reddit_partition_sizes(24, seed=1000, q=B.Q)[:12]
    parts = S.reddit_partition_stream(num_parts=6, seed=1000, nfeat=8, ncls=3, n=50, e_lo=100, e_hi=400, q=200, only={1, 4})
    assert [p is not None for p in parts] == [False, True, False, False, True, False]


def test_bench_self_launcher_starts_ranks_and_propagates_failure():
    """`python bench.py --gpus 2` with WORLD_SIZE unset starts its own rank processes (no GPU call in the parent, nothing
    re-exec'ed).  In the build container there is no GPU: every rank exits with bench.py's "needs an MI355X" message, and the parent
    must come back promptly with a non-zero status instead of a JSON line."""
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SGS_BENCH_REHEARSE="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    if torch.cuda.is_available():
        pytest.skip("failure propagation is checked where the ranks cannot find a GPU")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "needs an MI355X" in r.stderr and r.stdout.strip() == ""
