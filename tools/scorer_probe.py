"""What bounds the scorer forward?  HIP-event timings of sgs_edge_score_fwd under altered inputs:
  random   : src/dst uniform at random (worst-case gathers)
  sorted   : row-sorted edge list of a synthetic partition (the real layout: runs of equal src)
  same_row : every edge is (0, 0) -> both endpoint gathers hit one cache line set (gather cost removed)
  nodrop   : p = 0 (epilogue without the dropout hash)
and for E rounded to a whole number of 768-workgroup rounds (tail effect removed)."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sgs_gnn_amd as S

dev = "cuda:0"
N, H = 1013, 256
g = torch.Generator(device=dev).manual_seed(0)
codes = torch.relu(torch.randn(N, H, device=dev, generator=g))
fc1 = torch.nn.Linear(2 * H, H).to(dev)
fc2 = torch.nn.Linear(H, 1).to(dev)
part = S.synthetic_graph(N, 351194, 602, 41, seed=5, device=dev)

def run(ei, p, reps=20):
    E = ei.shape[1]
    with torch.no_grad():
        f = lambda: S.ops.edge_score(codes, fc1.weight, fc1.bias, fc2.weight, fc2.bias, ei, p=p, seed=1, site=2)
        f(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            f()
        b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    return {"E": E, "ms": round(ms, 4), "tflops": round(E * (2 * H * H + 2 * H) / ms / 1e9, 1)}

# warm the clocks / caches before any timing
S._lib.lib().sgs_edge_score_set_variant(1)
for _ in range(3):
    run(part.edge_index, 0.3, reps=50)
out = {}
for variant in (1, 3, 0, 2):
    S._lib.lib().sgs_edge_score_set_variant(variant)
    E = part.edge_index.shape[1]
    rnd = torch.randint(0, N, (2, E), device=dev, generator=g)
    r = {}
    r["sorted"] = run(part.edge_index, 0.3)
    r["random"] = run(rnd, 0.3)
    r["same_row"] = run(torch.zeros_like(rnd), 0.3)
    r["sorted_nodrop"] = run(part.edge_index, 0.0)
    Efull = (E // (768 * 64)) * 768 * 64
    r["sorted_whole_rounds"] = run(part.edge_index[:, :Efull].contiguous(), 0.3)
    r["same_row_nodrop_whole_rounds"] = run(torch.zeros(2, Efull, dtype=torch.int64, device=dev), 0.0)
    out[f"variant{variant}"] = r
print(json.dumps(out, indent=1))
