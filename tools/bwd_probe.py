"""A/B of the scorer backward core: LDS-tiled (0) vs 64-edge streaming loop (3) vs bf16x6 loop (4), by active-set size."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sgs_gnn_amd as S
ops = S.ops
L = S._lib.lib()
DEV = "cuda:0"
N, H, E = 1013, 256, 500000
g = torch.Generator().manual_seed(0)
codes = torch.relu(torch.randn(N, H, generator=g)).to(DEV)
ei = torch.randint(0, N, (2, E), generator=g).to(DEV)
W1 = (torch.randn(H, 2 * H, generator=g) / (2 * H) ** 0.5).to(DEV)
b1, w2, b2 = torch.zeros(H, device=DEV), (torch.randn(H, generator=g) / H ** 0.5).to(DEV), torch.zeros(1, device=DEV)
U = (codes @ W1[:, H:].t()).contiguous()
tile = L.sgs_edge_score_bwd_tile()
out = {}
for n in (100000, 262144, 500000):
    eid = torch.sort(torch.randperm(E, generator=g)[:n]).values.to(DEV)
    gp = torch.randn(n, generator=g).to(DEV)
    dv, feat, dz = torch.empty(n, H, device=DEV), torch.empty(n, H, device=DEV), torch.empty(n, device=DEV)
    hdz = torch.empty((n + tile - 1) // tile, H, device=DEV)
    ws = ops.workspace(L.sgs_edge_score_workspace_bytes(N, H, 0), codes.device)
    for variant in (0, 3, 4):
        L.sgs_edge_score_set_bwd_variant(variant)
        def f():
            S._lib.check(L.sgs_edge_score_bwd_core(codes.data_ptr(), U.data_ptr(), N, H, ei.data_ptr(), E, 0, eid.data_ptr(), n, gp.data_ptr(),
                                                   W1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), 0.3, 5, 2, dv.data_ptr(), hdz.data_ptr(),
                                                   dz.data_ptr(), feat.data_ptr(), ws.data_ptr(), ws.numel(), ops._stream()), "bwd_core")
        for _ in range(3): f()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): f()
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) / 10 * 1e3
        out[f"n={n} variant={variant}"] = {"us": round(us, 1), "tflops_recompute": round(n * (2 * H * H + 2 * H) / us / 1e6, 1)}
L.sgs_edge_score_set_bwd_variant(-1)
print(json.dumps(out, indent=1))
