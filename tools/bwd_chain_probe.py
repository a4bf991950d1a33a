"""Per-entry-point HIP-event times of the scorer backward's two forms at a Reddit partition's shape (n = 1 013, H = 256, q = 100 000 active
rows drawn from a row-sorted 350 k-edge partition): unfused (prep + dfeat_bits + gemm_tn_mask + endpoint_reduce_pair_bits) against fused
(prep_sd + dfeat_fused + gemm_tn_mask_gather + reduce_fused).  `python tools/bwd_chain_probe.py [q]`."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sgs_gnn_amd as S
ops = S.ops
L = S._lib.lib()
DEV = "cuda:0"
N, H, F_, C = 1013, 256, 602, 41
q = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
b = S.synthetic_graph(N, 350_000, F_, C, seed=71, device=DEV)
E = b.edge_index.shape[1]
g = torch.Generator(device=DEV).manual_seed(0)
eid = torch.sort(torch.randperm(E, device=DEV, generator=g)[:q]).values
sub = b.edge_index[:, eid].contiguous()
graph = ops.Graph(sub, N)
codes = torch.relu(torch.randn(N, H, device=DEV, generator=g))
W1 = torch.randn(H, 2 * H, device=DEV, generator=g) / (2 * H) ** 0.5
w2 = torch.randn(H, device=DEV, generator=g) / H ** 0.5
gp = torch.randn(q, device=DEV, generator=g)
p_out = torch.rand(E, device=DEV, generator=g)
maskbits = torch.randint(-2**31, 2**31 - 1, (E, H // 32), device=DEV, generator=g, dtype=torch.int64).to(torch.int32)
p = 0.3
f32 = dict(dtype=torch.float32, device=DEV)
dz, bits, feat, dfeat = torch.empty(q, **f32), torch.empty(q, H // 32, dtype=torch.int32, device=DEV), torch.empty(q, H, **f32), torch.empty(q, H, **f32)
sd = torch.empty(q, 2, dtype=torch.int32, device=DEV)
G, opart = torch.empty(q, H, **f32), torch.empty(L.sgs_edge_score_bwd_fused_opart_rows(q, N), H, **f32)
dW1, db1, db2, Traw, craw = torch.empty(H, 2 * H, **f32), torch.empty(H, **f32), torch.empty(1, **f32), torch.empty(H, H, **f32), torch.empty(H, **f32)
dcodes, dU, Rraw = torch.empty(N, H, **f32), torch.empty(N, H, **f32), torch.empty(N, H, **f32)
wsd = ops.workspace(max(L.sgs_edge_score_workspace_bytes(0, H, 0), L.sgs_gemm_tn_workspace_bytes(q, H, H)) * 2, codes.device)
st = ops._stream()
ck = S._lib.check
scale = 1.0 / (1.0 - p)
calls = {
    "prep (dz, bits, feat)": lambda: ck(L.sgs_edge_score_bwd_prep(codes.data_ptr(), N, H, b.edge_index.data_ptr(), E, eid.data_ptr(), q, gp.data_ptr(), p_out.data_ptr(),
                                                                  maskbits.data_ptr(), dz.data_ptr(), bits.data_ptr(), feat.data_ptr(), st)),
    "dfeat_bits": lambda: ck(L.sgs_edge_score_bwd_dfeat_bits(bits.data_ptr(), dz.data_ptr(), q, H, W1.data_ptr(), w2.data_ptr(), p, dfeat.data_ptr(), wsd.data_ptr(),
                                                             wsd.numel(), st)),
    "gemm_tn_mask": lambda: ck(L.sgs_gemm_tn_mask(bits.data_ptr(), dz.data_ptr(), w2.data_ptr(), scale, feat.data_ptr(), q, H, H, dW1.data_ptr(), 2 * H, db1.data_ptr(),
                                                  db2.data_ptr(), Traw.data_ptr(), craw.data_ptr(), wsd.data_ptr(), wsd.numel(), st)),
    "endpoint_reduce_pair_bits": lambda: ck(L.sgs_endpoint_reduce_pair_bits(dfeat.data_ptr(), bits.data_ptr(), dz.data_ptr(), w2.data_ptr(), p, codes.data_ptr(), N, H, q,
                                                                            graph.in_ptr.data_ptr(), graph.in_src.data_ptr(), graph.in_eid.data_ptr(),
                                                                            graph.out_ptr.data_ptr(), graph.out_dst.data_ptr(), graph.out_eid.data_ptr(),
                                                                            dcodes.data_ptr(), dU.data_ptr(), Rraw.data_ptr(), st)),
    "FUSED prep_sd (dz, bits, sd)": lambda: ck(L.sgs_edge_score_bwd_prep_sd(codes.data_ptr(), N, H, b.edge_index.data_ptr(), E, eid.data_ptr(), q, gp.data_ptr(),
                                                                            p_out.data_ptr(), maskbits.data_ptr(), dz.data_ptr(), bits.data_ptr(), sd.data_ptr(), st)),
    "FUSED dfeat_fused": lambda: ck(L.sgs_edge_score_bwd_dfeat_fused(bits.data_ptr(), dz.data_ptr(), sd.data_ptr(), codes.data_ptr(), q, N, H, W1.data_ptr(), w2.data_ptr(), p,
                                                                     G.data_ptr(), opart.data_ptr(), wsd.data_ptr(), wsd.numel(), st)),
    "FUSED gemm_tn_mask_gather": lambda: ck(L.sgs_gemm_tn_mask_gather(bits.data_ptr(), dz.data_ptr(), w2.data_ptr(), scale, codes.data_ptr(), N, sd.data_ptr(), q, H, H,
                                                                      dW1.data_ptr(), 2 * H, db1.data_ptr(), db2.data_ptr(), Traw.data_ptr(), craw.data_ptr(),
                                                                      wsd.data_ptr(), wsd.numel(), st)),
    "FUSED reduce_fused": lambda: ck(L.sgs_edge_score_bwd_reduce_fused(G.data_ptr(), opart.data_ptr(), bits.data_ptr(), dz.data_ptr(), w2.data_ptr(), p, N, H, q,
                                                                       graph.in_ptr.data_ptr(), graph.in_eid.data_ptr(), graph.out_ptr.data_ptr(), dcodes.data_ptr(),
                                                                       dU.data_ptr(), Rraw.data_ptr(), st)),
}
out = {"q": q, "E": E}
for name, f in calls.items():
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        f()
    e.record()
    torch.cuda.synchronize()
    out[name] = round(a.elapsed_time(e) / 20 * 1e3, 1)
out["unfused_total_us"] = round(sum(v for k, v in out.items() if k not in ("q", "E") and not k.startswith("FUSED")), 1)
out["fused_total_us"] = round(sum(v for k, v in out.items() if k.startswith("FUSED")), 1)
print(json.dumps(out, indent=1))
