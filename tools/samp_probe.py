"""HIP-event time of one learned draw (ops.sample_topq, fused small-E path) at the bench partition's size: python tools/samp_probe.py"""
import sys, torch
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), ".."))
import sgs_gnn_amd as S
from sgs_gnn_amd import ops
dev = torch.device("cuda:0")
b = S.synthetic_graph(1013, 351194, 16, 5, seed=3, train_frac=0.5, device=dev)
E = b.edge_index.shape[1]
p = torch.rand(E, device=dev)
for _ in range(3): ops.sample_topq(ops.SAMPLE_LEARNED, p, b.prob, 0.3, 100000, b.edge_index, seed=1, stream_id=2)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(50): ops.sample_topq(ops.SAMPLE_LEARNED, p, b.prob, 0.3, 100000, b.edge_index, seed=1, stream_id=2 + i)
e1.record(); torch.cuda.synchronize()
print("E", E, "us per draw", e0.elapsed_time(e1) / 50 * 1e3)
