"""cProfile of the host side of the benchmark steps (run on the GPU box)."""
import cProfile, pstats, io, os, sys, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
import sgs_gnn_amd as S

dev = "cuda:0"
S.fix_seeds(42)
model, og, oe, oa = B.build_model(S, dev)
args = B.make_args(dev)
pool = S.reddit_partition_stream(num_parts=12, seed=1000, nfeat=B.NFEAT, ncls=B.NCLS, n=B.N_NODES, q=B.Q, device=dev)
crit = torch.nn.CrossEntropyLoss()
with contextlib.redirect_stdout(io.StringIO()):
    S.train(args, 0, 10, model, og, oe, oa, crit, pool[:6], q=B.Q)
torch.cuda.synchronize()
timed = [pool[i % 12] for i in range(48)]
pr = cProfile.Profile()
pr.enable()
S.train(args, 1, 10, model, og, oe, oa, crit, timed, q=B.Q)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
