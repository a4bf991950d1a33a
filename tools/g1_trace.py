"""Replay the captured segments (slot graphs) with a big and a small partition staged, a few times each (run under
rocprofv3 --kernel-trace; analyse with g1_trace_analyze.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
import sgs_gnn_amd as S

dev = "cuda:0"
S.fix_seeds(42)
model, og, oe, oa = B.build_model(S, dev, fused=True)
crit = torch.nn.CrossEntropyLoss()
args = B.make_args(dev)
args.sgs_hipgraph = True
pool = S.reddit_partition_stream(num_parts=int(os.environ.get("PARTS", "24")), seed=1000, nfeat=B.NFEAT, ncls=B.NCLS, n=B.N_NODES, q=B.Q, device=dev)
target = int(os.environ.get("TARGET_E", "350000"))
big = min((b for b in pool if b.edge_index.shape[1] > B.Q), key=lambda b: abs(b.edge_index.shape[1] - target))
small = min(pool, key=lambda b: b.edge_index.shape[1])
import contextlib, io
with contextlib.redirect_stdout(io.StringIO()):
    S.prepare_step_graphs(args, model, og, oe, crit, pool, q=B.Q)
    for ep in range(2):
        S.train(args, ep, 10, model, og, oe, oa, crit, [big, small], q=B.Q)
sg = model._sgs_stepgraphs
S.ops.set_rng_epoch_buffer(sg.epoch_word)
c = sg.slots[True][0]
cs = sg.slots[False][0]
sg._stage(big, c)
sg._stage(small, cs)
torch.cuda.synchronize()
print("E_big", big.edge_index.shape[1], "E_small", small.edge_index.shape[1])
for g in ([c.g0] if c.g0 is not None else []) + [c.g1, c.g2r, c.g2l, cs.g1]:
    for _ in range(3):
        torch.cuda.synchronize()
        time.sleep(0.02)        # replays are separated by > 10 ms of idle time in the trace
        g.replay()
        torch.cuda.synchronize()
# HIP-event timing of the same segments (no profiler gaps)
def t(g, reps=20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.replay(); torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        g.replay()
    b.record(); torch.cuda.synchronize()
    return round(a.elapsed_time(b) / reps * 1e3, 1)
time.sleep(0.05)
print({"g0_us": t(c.g0) if c.g0 is not None else None, "g1_us": t(c.g1), "g2l_us": t(c.g2l), "g2r_us": t(c.g2r), "g_unsampled_us": t(cs.g1)})
print("done")
