import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import sgs_gnn_amd as S
import test_gpu_edge_score as T
ops = S.ops
L = S._lib.lib()
DEV = "cuda:0"
N, H, p = 777, 256, 0.0
E, q = 140_000, 66_000
codes, ei, W1, b1, W2, b2, g = T._case(N, H, E, 99)
eid = torch.sort(torch.randperm(E, generator=g)[:q]).values
gp = torch.zeros(E); gp[eid] = torch.randn(q, generator=g)
sub = ei[:, eid]
# forward bits vs oracle on this very case
d = lambda t: t.to(DEV).contiguous()
codes_d, ei_d, W1_d, b1_d, w2_d, b2_d = d(codes), d(ei), d(W1), d(b1), d(W2.reshape(-1)), d(b2)
U_d = (codes_d @ W1_d[:, H:].t()).contiguous()
ws = ops.workspace(L.sgs_edge_score_workspace_bytes(N, H, E), codes_d.device)
pm = torch.empty(E, device=DEV); bits = torch.zeros(E, H // 32, dtype=torch.int32, device=DEV)
S._lib.check(L.sgs_edge_score_fwd_mask(codes_d.data_ptr(), U_d.data_ptr(), N, H, ei_d.data_ptr(), E, 0, None, 0, None, W1_d.data_ptr(), b1_d.data_ptr(),
                                       w2_d.data_ptr(), b2_d.data_ptr(), p, 5, 2, pm.data_ptr(), bits.data_ptr(), ws.data_ptr(), ws.numel(), ops._stream()), "fwd_mask")
x, y = codes.double()[ei[0]], codes.double()[ei[1]]
v = torch.cat([x * y, x - y], 1) @ W1.double().t() + b1.double()
want = v > 0
got = ((bits.cpu().view(E, H // 32, 1) >> torch.arange(32).view(1, 1, 32)) & 1).bool().view(E, H)
wrong = (got != want) & (v.abs() > 1e-5)
print("forward bits wrong:", int(wrong.sum()), "rows", torch.nonzero(wrong.sum(1)).flatten()[:10].tolist(), "cols", torch.nonzero(wrong.sum(0)).flatten()[:20].tolist())
grads = {}
for form, (fwd_mask, mask) in {"kept": (True, True), "bits": (False, True), "dense": (False, False)}.items():
    ops._fwd_mask, ops._mask_backward = fwd_mask, mask
    dl = [t.clone().to(DEV).requires_grad_(True) for t in (codes, W1, b1, W2, b2)]
    act = ops.ActiveSet()
    pd = ops.edge_score(dl[0], dl[1], dl[2], dl[3], dl[4], ei.to(DEV), active=act, p=p, seed=5, site=2)
    act.set(eid.to(DEV), ops.Graph(sub.to(DEV), N))
    pd.backward(gp.to(DEV))
    grads[form] = [t.grad.detach().cpu() for t in dl]
ops._fwd_mask, ops._mask_backward = True, True
for form in ("kept", "bits"):
    print(form, {n: f"{T._rel(a, b):.2e}" for n, a, b in zip(["dcodes", "dW1", "db1", "dW2", "db2"], grads[form], grads["dense"])})
# the autograd path's kept mask
ops._fwd_mask, ops._mask_backward = True, True
dl = [t.clone().to(DEV).requires_grad_(True) for t in (codes, W1, b1, W2, b2)]
act = ops.ActiveSet()
pd = ops.edge_score(dl[0], dl[1], dl[2], dl[3], dl[4], ei.to(DEV), active=act, p=p, seed=5, site=2)
st = pd.grad_fn.saved_tensors
print("saved:", [tuple(t.shape) for t in st])
mb = st[7]
got2 = ((mb.cpu().view(E, H // 32, 1) >> torch.arange(32).view(1, 1, 32)) & 1).bool().view(E, H)
wrong2 = (got2 != want) & (v.abs() > 1e-5)
print("autograd-path kept bits wrong:", int(wrong2.sum()), "rows", torch.nonzero(wrong2.sum(1)).flatten()[:10].tolist(), "cols", torch.nonzero(wrong2.sum(0)).flatten()[:24].tolist())
print("equal to the direct call's bits:", bool(torch.equal(mb, bits)), " p equal:", bool(torch.equal(st[8], pm)))
# fp64 oracle for the same active rows
from oracle import sgs_oracle as O
Ao = codes.clone().double().requires_grad_(True)
Po = [t.clone().double().requires_grad_(True) for t in (W1, b1, W2, b2)]
po = O.edge_score(Ao[ei[0, eid]], Ao[ei[1, eid]], Po[0], Po[1], Po[2], Po[3]).squeeze(1)
po.backward(gp[eid].double())
ref = [Ao.grad, Po[0].grad, Po[1].grad, Po[2].grad.reshape(-1), Po[3].grad]
for form in ("kept", "bits", "dense"):
    print(form, "vs fp64 oracle:", {n: f"{T._rel(a.reshape(b.shape), b):.2e}" for n, a, b in zip(["dcodes", "dW1", "db1", "dW2", "db2"], grads[form], ref)})
pc = pm.cpu().double()[eid]
dzr = gp[eid].double() * pc * (1 - pc)
for name, mask in (("mask from fp64 v>0", want[eid]), ("kept mask", got2[eid])):
    c = (mask.double() * dzr[:, None]).sum(0)
    db1_ref = c * W2.double().reshape(-1)
    print(name, "-> db1: kept", f"{T._rel(grads['kept'][2], db1_ref):.2e}", "bits", f"{T._rel(grads['bits'][2], db1_ref):.2e}", "dense", f"{T._rel(grads['dense'][2], db1_ref):.2e}")
print("kept mask vs fp64 mask on active rows: differing bits", int((want[eid] != got2[eid]).sum()), " with |v| > 1e-5:", int(((want[eid] != got2[eid]) & (v[eid].abs() > 1e-5)).sum()))
