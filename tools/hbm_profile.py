import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sgs_gnn_amd as S
ops = S.ops
dev = "cuda:0"
N, E, q = 232965, 114615892, 22923178
g = torch.Generator(device=dev).manual_seed(0)
p = torch.rand(E, device=dev, generator=g)
prior = torch.softmax(torch.rand(E, device=dev, generator=g), 0)
ei = torch.randint(0, N, (2, E), device=dev, generator=g)
ei = ei[:, torch.argsort(ei[0] * N + ei[1])].contiguous()
for _ in range(2):
    smp = ops.sample_topq(ops.SAMPLE_LEARNED, p, prior, 0.3, q, ei, seed=1, stream_id=1, want_p=True)
del p, prior
for _ in range(2):
    gr = ops.Graph(smp.edge_index, N)
w = torch.rand(q, device=dev, generator=g)
for _ in range(2):
    nm = ops.gcn_norm(gr, w)
torch.cuda.synchronize()
print("ok")
