"""Same-process A/B of host-side choices of the graph-mode training loop on the bench stream: whole-epoch wall time (230 shuffled partitions,
steady gate mix) with a switch off / on, alternating.  `python tools/host_ab.py`"""
import contextlib, io, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
import sgs_gnn_amd as S
from sgs_gnn_amd import training as T
dev = "cuda:0"
S.fix_seeds(42)
model, og, oe, oa = B.build_model(S, dev, fused=True)
crit = torch.nn.CrossEntropyLoss()
args = B.make_args(dev); args.sgs_hipgraph = True
pool = S.reddit_partition_stream(num_parts=230, seed=1000, nfeat=B.NFEAT, ncls=B.NCLS, n=B.N_NODES, q=B.Q, device=dev)
with contextlib.redirect_stdout(io.StringIO()):
    S.prepare_step_graphs(args, model, og, oe, crit, pool, q=B.Q)
    for ep in range(2):
        S.train(args, ep, 10, model, og, oe, oa, crit, pool, q=B.Q)
def epoch(ep):
    torch.cuda.synchronize(); t = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        S.train(args, ep, 10, model, og, oe, oa, crit, pool, q=B.Q)
    torch.cuda.synchronize(); return time.perf_counter() - t
out = {"zero_grad every step": [], "zero_grad only for eager steps": []}
for r in range(4):
    T._ALWAYS_ZERO_GRAD = True
    out["zero_grad every step"].append(round(epoch(10 + 2 * r) * 1e3, 2))
    T._ALWAYS_ZERO_GRAD = False
    out["zero_grad only for eager steps"].append(round(epoch(11 + 2 * r) * 1e3, 2))
print(json.dumps(out, indent=1))
