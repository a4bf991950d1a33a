"""Bisect probe: the sampled hybrid forward captured as five chained graphs, sync + progress line after each."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sgs_gnn_amd as S
from sgs_gnn_amd import ops
from sgs_gnn_amd.sampling import draw_learned, draw_prior
from sgs_gnn_amd.training import sampled_forward, learned_loss, _ce, SampledForward

DEV = "cuda:0"
MODE = sys.argv[1] if len(sys.argv) > 1 else "split"
a = argparse.Namespace(device=DEV, mode="learned", pipeline="hybrid", edge_mlp_type="GCN", conditional=True,
                       sparse_edge_mlp=True, t_init=0.7, t_min=0.5, degree_bias_coef=0.3, reg1=True, reg2=True,
                       regularizer1_coef=1.0, consist_reg_coef=0.5, hybrid_checkpoint=False, drop_rate=0.0, lr=1e-2)
torch.manual_seed(3); S.fix_seeds(3)
m = S.GNNModel(24, 32, 5, dropout_prob=0.0, edge_mlp_type="GCN").to(DEV)
og = torch.optim.Adam([p_ for n, p_ in m.named_parameters() if "gcn" in n], lr=1e-2)
oe = torch.optim.Adam([p_ for n, p_ in m.named_parameters() if "edge_prob_mlp" in n], lr=1e-2)
crit = torch.nn.CrossEntropyLoss()
b = S.synthetic_graph(120, 4000, 24, 5, seed=11, device=DEV)
q, N = 800, 120
epoch_word = torch.zeros(1, dtype=torch.int64, device=DEV)
ops.set_rng_epoch_buffer(epoch_word)
ops.pin_workspaces(True)
side = torch.cuda.Stream()

def eager_step(opt=True):
    for p_ in m.parameters():
        p_.grad = None
    st = sampled_forward("hybrid", a, m, b, q, False)
    loss = learned_loss(a, crit, st, b)
    loss.backward()
    if opt:
        oe.step(); og.step()

# warm-up on the side stream (both branches)
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    st = sampled_forward("hybrid", a, m, b, q, False)
    _ce(crit, st.random_out, b).backward(retain_graph=True)
    learned_loss(a, crit, st, b).backward()
    for p_ in m.parameters():
        p_.grad = None
    del st
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
for mod in m.modules():
    if hasattr(mod, "_lin_cache"):
        mod._lin_cache = None

graphs, names = [], []
class _NS(SampledForward):
    pass
st = _NS()
sc = m.edge_prob_mlp
pool = None
def cap(name, fn):
    global pool
    g = torch.cuda.CUDAGraph()
    kw = {} if pool is None else {"pool": pool}
    with torch.cuda.graph(g, stream=side, **kw):
        fn()
    if pool is None:
        pool = g.pool()
    graphs.append(g); names.append(name)

def fA():
    epoch_word.add_(1)
    st.rs = draw_prior(b.prob, b.edge_index, q)
    st.rsei = st.rs.edge_index
def fB():
    st.edge_probs_full = sc(b.x, b.edge_index, st.rsei).squeeze()
def fC():
    st.smp = draw_learned(b.prob, st.edge_probs_full, b.edge_index, q, a.degree_bias_coef)
    st.sampled_edge_index = st.smp.edge_index
def fD1():
    st.graph_s = ops.get_graph(st.smp.edge_index, N)
def fD2():
    sc.last_active.set(st.smp.eid, st.graph_s)
    st.edge_probs_for_loss = st.edge_probs_full.index_select(0, st.smp.eid)
    st.learned_out = m(b, st.smp.edge_index, st.edge_probs_for_loss)
def fE():
    st.random_out = m(b, st.rsei)
    st.cbuf = torch.empty(5, dtype=torch.int32, device=DEV)
    ops.masked_correct(st.learned_out, b.y, b.train_mask, out=st.cbuf[0:2])
    ops.masked_correct(st.random_out, b.y, b.train_mask, out=st.cbuf[2:4])
if MODE == "split":
    for nme, f in (("A prior", fA), ("B score", fB), ("C learned draw", fC), ("D1 csr", fD1), ("D2 learned enc", fD2), ("E random enc", fE)):
        cap(nme, f)
else:
    cap("G1", lambda: (fA(), fB(), fC(), fD1(), fD2(), fE()))
def fL():
    st.loss_l = learned_loss(a, crit, st, b)
    st.loss_l.backward(retain_graph=True)
cap("G2L", fL)
for p_ in m.parameters():
    p_.grad = None
def fR():
    st.loss_r = _ce(crit, st.random_out, b)
    st.loss_r.backward()
cap("G2R", fR)
for p_ in m.parameters():
    p_.grad = None
print("captured", names, flush=True)

for it in range(8):
    for g, nme in zip(graphs, names):
        g.replay()
        torch.cuda.synchronize()
        extra = ""
        if nme.startswith("C") or nme == "G1":
            e = st.smp.eid
            extra = f"eid[{int(e.min())},{int(e.max())}] sorted={bool((e[1:] > e[:-1]).all())} sei[{int(st.smp.edge_index.min())},{int(st.smp.edge_index.max())}]"
        if nme.startswith("A"):
            extra = f"rsei[{int(st.rsei.min())},{int(st.rsei.max())}]"
        if nme.startswith("D1"):
            gs = st.graph_s
            extra = f"in_ptr[{int(gs.in_ptr.min())},{int(gs.in_ptr.max())}] in_src[{int(gs.in_src.min())},{int(gs.in_src.max())}] in_eid[{int(gs.in_eid.min())},{int(gs.in_eid.max())}] out_dst[{int(gs.out_dst.min())},{int(gs.out_dst.max())}] maxdeg={int((gs.in_ptr[1:]-gs.in_ptr[:-1]).max())}"
        print(it, nme, "ok", extra, flush=True)
    eager_step(); eager_step(); eager_step()
    torch.cuda.synchronize()
    print(it, "eager ok", flush=True)
print("OK", MODE, flush=True)
