"""The bandwidth-bound kernels at full-graph scale (BASELINE config 5: Reddit, E = 114.6 M candidate edges, q = 22.9 M, N = 232 965):
achieved GB/s from HIP events against the HBM roofline (8 TB/s spec, ~4.7 TB/s measured device copy).  Algorithmic bytes per
SURVEY.md section 8d."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sgs_gnn_amd as S
ops = S.ops
L = S._lib.lib()
dev = "cuda:0"
N, E, q = 232965, 114615892, 22923178
g = torch.Generator(device=dev).manual_seed(0)

def timeit(f, reps=5):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

out = {"config": {"N": N, "E": E, "q": q}, "hbm_peak_GBps": 8000, "measured_copy_GBps": 4700}
# ---- sampler (learned draw, in-kernel noise)
p = torch.rand(E, device=dev, generator=g)
prior = torch.softmax(torch.rand(E, device=dev, generator=g), 0)
ei = torch.randint(0, N, (2, E), device=dev, generator=g)
ei = ei[:, torch.argsort(ei[0] * N + ei[1])].contiguous()
r = {}
ms = timeit(lambda: r.__setitem__("s", ops.sample_topq(ops.SAMPLE_LEARNED, p, prior, 0.3, q, ei, seed=1, stream_id=1, want_p=True)), 3)
by = 8 * E + 36 * q
out["sampler_learned_draw"] = {"ms": round(ms, 3), "algorithmic_bytes": by, "GBps": round(by / ms / 1e6, 1), "frac_of_8TBps": round(by / ms / 1e6 / 8000, 3),
                               "note": "8 B/candidate (p, prior) + 36 B/selected (i64 index pair read + written, weight); the kernels also write and re-read "
                                       "4 B keys per candidate in 3 digit passes (not counted as algorithmic)"}
smp = r["s"]
del p, prior
torch.cuda.empty_cache()
# ---- CSR build of the sampled graph
sei = smp.edge_index
ms = timeit(lambda: ops.Graph(sei, N), 3)
by = 16 * q
out["csr_build_both_orientations"] = {"ms": round(ms, 3), "algorithmic_bytes": by, "GBps": round(by / ms / 1e6, 1), "frac_of_8TBps": round(by / ms / 1e6 / 8000, 3),
                                      "note": "16 B per edge read (i64 pair); produces 2 x (ptr, col, eid) i32 + per-row sort"}
parent = ops.get_graph(ei, N)                      # the full graph's CSR: built once, cached
torch.cuda.synchronize()
ms = timeit(lambda: ops.get_subgraph(ei, N, smp), 3)
by = 2 * 4 * E + E + 16 * q
out["csr_of_drawn_subgraph_by_filter"] = {"ms": round(ms, 3), "algorithmic_bytes": by, "GBps": round(by / ms / 1e6, 1), "frac_of_8TBps": round(by / ms / 1e6 / 8000, 3),
                                         "note": "two passes over the parent's 2 x E edge-id entries + mask; writes 2 x (col, eid) i32 per selected edge"}
gr = ops.get_graph(sei, N)
w = torch.rand(q, device=dev, generator=g)
ms = timeit(lambda: ops.gcn_norm(gr, w), 5)
by = q * (4 + 4 + 4) * 2 + N * 16
out["gcn_norm_fwd"] = {"ms": round(ms, 3), "algorithmic_bytes": by, "GBps": round(by / ms / 1e6, 1), "frac_of_8TBps": round(by / ms / 1e6 / 8000, 3)}
nm = ops.gcn_norm(gr, w)
for D in (256, 41):
    X = torch.randn(N, D, device=dev, generator=g)
    ms = timeit(lambda: ops.gcn_propagate(X, nm, None, ops.ACT_RELU), 5)
    nnz = q
    upper = nnz * (4 + 4 + 4 * D) + 4 * D * N * 2
    comp = nnz * 8 + 4 * D * N * 2
    out[f"spmm_D{D}"] = {"ms": round(ms, 3), "bytes_uncached_gathers": upper, "bytes_compulsory": comp, "GBps_uncached": round(upper / ms / 1e6, 1),
                         "GBps_compulsory": round(comp / ms / 1e6, 1), "frac_of_8TBps_uncached": round(upper / ms / 1e6 / 8000, 3),
                         "note": "uncached = every gathered row counted (4 D B per nnz); compulsory = X read once + Y written + indices"}
print(json.dumps(out, indent=1))
