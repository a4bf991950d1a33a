"""HIP-event time of sgs_gemm_tn_mask_gather (the scorer's weight gradient d W1a = dv^T (codes[src] * codes[dst]) with dv as mask bits) at the
bench's shape (K = q = 100 000 rows, M = N = 256, 1 013-row table): per-wave-slice kernel of round 2 against the shared-operand kernel, by
number of K-slabs.  `python tools/gather_gemm_probe.py [K] [H] [N]`."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sgs_gnn_amd as S
ops, L = S.ops, S._lib.lib()
DEV = "cuda:0"
K = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
H = int(sys.argv[2]) if len(sys.argv) > 2 else 256
N = int(sys.argv[3]) if len(sys.argv) > 3 else 1013
g = torch.Generator(device=DEV).manual_seed(0)
codes = torch.relu(torch.randn(N, H, device=DEV, generator=g))
sd = torch.sort(torch.randint(0, N, (K, 2), device=DEV, generator=g, dtype=torch.int32), dim=0).values.contiguous()
bits = torch.randint(-2**31, 2**31 - 1, (K, H // 32), device=DEV, generator=g, dtype=torch.int64).to(torch.int32)
dz, w2 = torch.randn(K, device=DEV, generator=g), torch.randn(H, device=DEV, generator=g)
ws = ops.workspace(L.sgs_gemm_tn_workspace_bytes(K, H, H), codes.device)
C = torch.empty(H, 2 * H, device=DEV)
cs, dzs, Craw, csraw = torch.empty(H, device=DEV), torch.empty(1, device=DEV), torch.empty(H, H, device=DEV), torch.empty(H, device=DEV)
def call():
    S._lib.check(L.sgs_gemm_tn_mask_gather(bits.data_ptr(), dz.data_ptr(), w2.data_ptr(), 1.0 / 0.7, codes.data_ptr(), N, sd.data_ptr(), K, H, H, C.data_ptr(),
                                           2 * H, cs.data_ptr(), dzs.data_ptr(), Craw.data_ptr(), csraw.data_ptr(), ws.data_ptr(), ws.numel(), ops._stream()), "g")
out = {"K": K, "H": H, "N": N}
for name, shared, slabs in (("per-wave slices (round 2)", 0, 0), ("shared, automatic", 1, 0), ("shared, 16 slabs", 1, 16), ("shared, 32 slabs", 1, 32),
                            ("shared, 48 slabs", 1, 48), ("shared, 64 slabs", 1, 64), ("shared, 96 slabs", 1, 96), ("shared, 128 slabs", 1, 128)):
    L.sgs_gemm_tn_set_gather_variant(shared, slabs)
    for _ in range(5): call()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50): call()
    b.record(); torch.cuda.synchronize()
    out[name] = round(a.elapsed_time(b) / 50 * 1e3, 1)
L.sgs_gemm_tn_set_gather_variant(1, 0)
print(json.dumps(out, indent=1))
