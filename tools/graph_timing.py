"""GPU time of the captured step segments (HIP events around graph replays), per partition of the bench pool."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
import sgs_gnn_amd as S
from sgs_gnn_amd.stepgraph import StepGraphs

dev = "cuda:0"
S.fix_seeds(42)
model, og, oe, oa = B.build_model(S, dev, fused=True)
crit = torch.nn.CrossEntropyLoss()
args = B.make_args(dev)
args.sgs_hipgraph = True
pool = S.reddit_partition_stream(num_parts=12, seed=1000, nfeat=B.NFEAT, ncls=B.NCLS, n=B.N_NODES, q=B.Q, device=dev)
import contextlib, io
with contextlib.redirect_stdout(io.StringIO()):
    for ep in range(3):
        S.train(args, ep, 10, model, og, oe, oa, crit, pool, q=B.Q)
sg = model._sgs_stepgraphs
print(json.dumps({"partitions_captured": len(sg.table), "hbm_allocated_GiB": round(torch.cuda.memory_allocated() / 2**30, 2), "hbm_reserved_GiB": round(torch.cuda.memory_reserved() / 2**30, 2)}))
S.ops.set_rng_epoch_buffer(sg.epoch_word)

def t(g, reps=20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.replay(); torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

import time
def host_us(g, reps=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return (t1 - t0) / reps * 1e6

rows = []
for bt in pool:
    from sgs_gnn_amd.stepgraph import _batch_key
    c = sg.table[_batch_key(bt)]
    E = bt.edge_index.shape[1]
    if c.sampled:
        rows.append(dict(E=E, g0_us=round(t(c.g0), 1) if c.g0 is not None else None, g1_us=round(t(c.g1), 1), g2l_us=round(t(c.g2l), 1), g2r_us=round(t(c.g2r), 1),
                         g1_host_launch_us=round(host_us(c.g1), 1), g2r_host_launch_us=round(host_us(c.g2r), 1)))
    else:
        rows.append(dict(E=E, g_us=round(t(c.g1), 1), g_host_launch_us=round(host_us(c.g1), 1)))
for r in sorted(rows, key=lambda r: r["E"]):
    print(json.dumps(r))
