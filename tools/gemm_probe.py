import torch, time
dev = "cuda:0"
q, H = 100000, 256
dv = torch.randn(q, H, device=dev)
W1 = torch.randn(H, 2 * H, device=dev)
feat = torch.randn(q, H, device=dev)
def t(f, reps=20):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
W1a_view = W1[:, :H]
W1a_c = W1a_view.contiguous()
W1aT_c = W1a_c.t().contiguous()
fl = 2.0 * q * H * H
for name, f in [("dv @ W1a (strided view)", lambda: dv @ W1a_view), ("dv @ W1a (contiguous)", lambda: dv @ W1a_c),
                ("dv @ (W1a^T contiguous).t()", lambda: dv @ W1aT_c.t()), ("(W1a^T @ dv^T)^T", lambda: (W1aT_c @ dv.t()).t()),
                ("dv.t() @ feat", lambda: dv.t() @ feat), ("linear(dv, W1aT)", lambda: torch.nn.functional.linear(dv, W1aT_c))]:
    us = t(f)
    print(f"{name:32s} {us:8.1f} us  {fl / us / 1e6:6.1f} TFLOP/s")
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sgs_gnn_amd as S
L = S._lib.lib()
C = torch.empty(H, H, device=dev)
ws = S.ops.workspace(L.sgs_gemm_tn_workspace_bytes(q, H, H), dv.device)
def own():
    S._lib.check(L.sgs_gemm_tn(dv.data_ptr(), feat.data_ptr(), q, H, H, C.data_ptr(), ws.data_ptr(), ws.numel(), S.ops._stream()), "gemm_tn")
for v, nm in ((0, "sgs_gemm_tn(dv, feat) fp32 MFMA"), (1, "sgs_gemm_tn(dv, feat) bf16x6")):
    L.sgs_gemm_tn_set_tall_variant(v)
    us = t(own)
    print(f"{nm:32s} {us:8.1f} us  {fl / us / 1e6:6.1f} TFLOP/s   max err vs torch {float((C - dv.t() @ feat).abs().max()):.3e}")
L.sgs_gemm_tn_set_tall_variant(-1)
