"""Scorer forward: time of each kernel variant on the bench's largest partition (HIP events, 20 launches)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
import sgs_gnn_amd as S

dev = "cuda:0"
S.fix_seeds(42)
model, *_ = B.build_model(S, dev, fused=True)
pool = S.reddit_partition_stream(num_parts=12, seed=1000, nfeat=B.NFEAT, ncls=B.NCLS, n=B.N_NODES, q=B.Q, device=dev)
big = max(pool, key=lambda b: b.edge_index.shape[1])
L = S._lib.lib()
out = {}
for v in [int(x) for x in (sys.argv[1:] or ["3", "4"])]:
    L.sgs_edge_score_set_variant(v)
    r = B.kernel_roofline(S, model, big, reps=20)
    out[v] = {"ms": r["ms_per_launch"], "tflops_fp32_equiv": r["achieved"]}
L.sgs_edge_score_set_variant(-1)
print(json.dumps(out))
