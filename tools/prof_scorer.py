"""Launch the dominant kernel (fused edge scorer, forward) a few times on a Reddit-partition-sized
input -- the target of `rocprofv3 --pmc ...` passes (profiles/README.md)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sgs_gnn_amd as S

dev = "cuda:0"
E = int(sys.argv[1]) if len(sys.argv) > 1 else 351194
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
N, H = 1013, 256
g = torch.Generator(device=dev).manual_seed(0)
codes = torch.relu(torch.randn(N, H, device=dev, generator=g))
ei = torch.randint(0, N, (2, E), device=dev, generator=g)
fc1 = torch.nn.Linear(2 * H, H).to(dev)
fc2 = torch.nn.Linear(H, 1).to(dev)
pairs = None
if len(sys.argv) > 3 and sys.argv[3] == "paired":
    # the paired entry point on the bench stream's largest partition (an undirected graph stored both ways): `E` is ignored
    sizes = S.reddit_partition_sizes(230, seed=1000, q=100_000)
    idx = max(range(len(sizes)), key=lambda i: sizes[i])
    ei = S.reddit_partition_stream(num_parts=230, seed=1000, nfeat=602, ncls=41, n=N, q=100_000, device=dev, only={idx})[idx].edge_index
    pairs = S.ops.get_pairs(ei, N, build=True)
    print("paired: E", ei.shape[1], "canonical", pairs[0].numel())
if pairs is not None:
    # the TRAINING forward (gradients enabled): the paired loop that also keeps the mask of every scored edge (sgs_edge_score_fwd_mask)
    codes.requires_grad_(True)
    for _ in range(reps):
        p = S.ops.edge_score(codes, fc1.weight, fc1.bias, fc2.weight, fc2.bias, ei, p=0.3, seed=1, site=2, pairs=pairs).detach()
else:
    with torch.no_grad():
        for _ in range(reps):
            p = S.ops.edge_score(codes, fc1.weight, fc1.bias, fc2.weight, fc2.bias, ei, p=0.3, seed=1, site=2, pairs=pairs)
torch.cuda.synchronize()
print("ok", float(p.mean()))
