"""cProfile of the graph-mode training loop (host side) over the bench pool."""
import os, sys, cProfile, pstats, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
import sgs_gnn_amd as S
dev = "cuda:0"
S.fix_seeds(42)
model, og, oe, oa = B.build_model(S, dev, fused=True)
crit = torch.nn.CrossEntropyLoss()
args = B.make_args(dev); args.sgs_hipgraph = True
pool = S.reddit_partition_stream(num_parts=12, seed=1000, nfeat=B.NFEAT, ncls=B.NCLS, n=B.N_NODES, q=B.Q, device=dev)
with contextlib.redirect_stdout(io.StringIO()):
    for ep in range(3):
        S.train(args, ep, 10, model, og, oe, oa, crit, pool, q=B.Q)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
with contextlib.redirect_stdout(io.StringIO()):
    for ep in range(5):
        S.train(args, 3 + ep, 10, model, og, oe, oa, crit, pool, q=B.Q)
torch.cuda.synchronize()
pr.disable()
dt = time.perf_counter() - t0
print(f"60 steps in {dt*1e3:.1f} ms (profiled)")
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(18); print(st.getvalue()[:4000])
