"""Replay the captured segments of the largest partition a few times (run under rocprofv3 --kernel-trace)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
import sgs_gnn_amd as S
from sgs_gnn_amd.stepgraph import _batch_key

dev = "cuda:0"
S.fix_seeds(42)
model, og, oe, oa = B.build_model(S, dev, fused=True)
crit = torch.nn.CrossEntropyLoss()
args = B.make_args(dev)
args.sgs_hipgraph = True
pool = S.reddit_partition_stream(num_parts=12, seed=1000, nfeat=B.NFEAT, ncls=B.NCLS, n=B.N_NODES, q=B.Q, device=dev)
big = max(pool, key=lambda b: b.edge_index.shape[1])
small = min(pool, key=lambda b: b.edge_index.shape[1])
import contextlib, io
with contextlib.redirect_stdout(io.StringIO()):
    for ep in range(3):
        S.train(args, ep, 10, model, og, oe, oa, crit, [big, small], q=B.Q)
sg = model._sgs_stepgraphs
S.ops.set_rng_epoch_buffer(sg.epoch_word)
c = sg.table[_batch_key(big)]
cs = sg.table[_batch_key(small)]
torch.cuda.synchronize()
import time
for g in ([c.g0] if c.g0 is not None else []) + [c.g1, c.g2r, c.g2l, cs.g1]:
    for _ in range(3):
        torch.cuda.synchronize()
        time.sleep(0.02)        # replays are separated by > 10 ms of idle time in the trace
        g.replay()
        torch.cuda.synchronize()
print("done")
