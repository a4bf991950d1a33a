"""HIP-event time of the scorer's training forward (paired + mask) and backward on the bench stream's largest partition for combinations of
the probe knobs: start-up stagger of the second resident workgroup of every CU (per kernel MODE), raised wave priority."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sgs_gnn_amd as S
ops = S.ops
L = S._lib.lib()
dev = "cuda:0"
N, H = 1013, 256
sizes = S.reddit_partition_sizes(230, seed=1000, q=100_000)
idx = max(range(len(sizes)), key=lambda i: sizes[i])
ei = S.reddit_partition_stream(num_parts=230, seed=1000, nfeat=602, ncls=41, n=N, q=100_000, device=dev, only={idx})[idx].edge_index
pairs = ops.get_pairs(ei, N, build=True)
g = torch.Generator(device=dev).manual_seed(0)
codes = torch.relu(torch.randn(N, H, device=dev, generator=g)).requires_grad_(True)
fc1 = torch.nn.Linear(2 * H, H).to(dev)
fc2 = torch.nn.Linear(H, 1).to(dev)
E = ei.shape[1]
eid = torch.sort(torch.randperm(E, device=dev, generator=g)[:100_000]).values
gp = torch.randn(100_000, device=dev, generator=g)

def fwd(reps):
    for _ in range(reps):
        p = ops.edge_score(codes, fc1.weight, fc1.bias, fc2.weight, fc2.bias, ei, p=0.3, seed=1, site=2, pairs=pairs)
    return p

def bwd(reps):
    p = fwd(1)
    sel = p[eid]
    for _ in range(reps):
        torch.autograd.grad(sel, [codes, fc1.weight], gp, retain_graph=True)

def timed(fn, reps):
    fn(5)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(reps); b.record(); torch.cuda.synchronize()
    return round(a.elapsed_time(b) / reps * 1e3, 1)

FWD = (1 << 0) | (1 << 3)
out = {}
for name, stagger, prio, mask in (("none", 0, 0, 0), ("policy", -1, 0, FWD), ("fwd 320", 320, 0, FWD), ("fwd 640", 640, 0, FWD),
                                   ("fwd 800", 800, 0, FWD), ("fwd 640 prio1", 640, 1, FWD), ("all modes 640", 640, 0, 0x3F),
                                   ("all modes 320", 320, 0, 0x3F)):
    S._lib.check(L.sgs_edge_score_probe_set(stagger, prio, mask))
    out[name] = {"forward_us": timed(fwd, 60), "backward_us": timed(bwd, 30)}
S._lib.check(L.sgs_edge_score_probe_set(-1, 0, FWD))
print(json.dumps(out, indent=1))
