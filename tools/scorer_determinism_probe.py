import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sgs_gnn_amd as S
ops = S.ops
DEV = "cuda:0"
n, H = 33869, 256
b = S.synthetic_graph(n, 463000, 128, 5, seed=300, train_frac=0.2, power=0.6, device=DEV)
g = torch.Generator(device=DEV).manual_seed(1)
codes = torch.relu(torch.randn(n, H, device=DEV, generator=g))
W1 = torch.randn(H, 2 * H, device=DEV, generator=g) / (2 * H) ** 0.5
b1 = torch.randn(H, device=DEV, generator=g) * 0.05
W2 = torch.randn(1, H, device=DEV, generator=g) / H ** 0.5
b2 = torch.zeros(1, device=DEV)
pairs = ops.get_pairs(b.edge_index, n, build=True)
with torch.no_grad():
    p0 = ops.edge_score(codes, W1, b1, W2, b2, b.edge_index, pairs=pairs).clone()
    bad = sum(int((ops.edge_score(codes, W1, b1, W2, b2, b.edge_index, pairs=pairs) != p0).any()) for _ in range(60))
print(os.environ.get("SGS_LIB_PATH", "default"), "-> nondeterministic runs:", bad, "/ 60")
