import sys, torch
sys.path.insert(0, '.')
import sgs_gnn_amd
from sgs_gnn_amd import ops
from oracle import sgs_oracle as O
DEV='cuda:0'
torch.manual_seed(0)
for loops in (False, True):
    N,E,D=30,200,8
    g=torch.Generator().manual_seed(1)
    ei=torch.randint(0,N,(2,E),generator=g)
    if not loops:
        keep=ei[0]!=ei[1]; ei=ei[:,keep]; E=ei.shape[1]
    else:
        ei[:,3]=ei[0,3]; ei[:,7]=ei[0,3]
    w=torch.rand(E,generator=g)
    X=torch.randn(N,D,generator=g)
    gy=torch.randn(N,D,generator=g)
    # oracle with explicit what
    wo=w.double().requires_grad_(True)
    ei2, what = O.gcn_norm(ei, wo, N, dtype=torch.float64)
    what.retain_grad()
    Xo=X.double().requires_grad_(True)
    out=torch.zeros(N,D,dtype=torch.float64).index_add(0, ei2[1], what.unsqueeze(1)*Xo[ei2[0]])
    out.backward(gy.double())
    # device
    gr=ops.Graph(ei.to(DEV),N)
    wd=w.to(DEV).requires_grad_(True)
    nm=ops.gcn_norm(gr,wd)
    Xd=X.to(DEV).requires_grad_(True)
    Y=ops.gcn_propagate(Xd,nm)
    nm.handle.retain_grad()
    Y.backward(gy.to(DEV))
    print("loops",loops,"fwd",float((Y.detach().cpu().double()-out.detach()).abs().max()))
    print(" dX", float((Xd.grad.cpu().double()-Xo.grad).abs().max()))
    hg=nm.handle.grad.cpu().double()
    # map oracle what.grad: non-loop edges in order then N loops
    mask=ei[0]!=ei[1]
    gw_o=torch.zeros(E,dtype=torch.float64); gw_o[mask]=what.grad[:int(mask.sum())]
    gl_o=what.grad[int(mask.sum()):]
    print(" gw", float((hg[:E][mask]-gw_o[mask]).abs().max()), " gloop", float((hg[E:]-gl_o).abs().max()))
    print(" dw", float((wd.grad.cpu().double()-wo.grad).abs().max()), float(wo.grad.abs().max()))
    print(" dis", float((nm.dis.cpu().double() - (torch.zeros(N,dtype=torch.float64).index_add(0, ei2[1], torch.cat([wo.detach()[mask], torch.ones(N,dtype=torch.float64)]) if not loops else torch.zeros(0))).pow(-0.5)).abs().max()) if not loops else "")
