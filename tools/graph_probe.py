"""Debug probe for HIP-graph capture of the sampled step: replays with a sync + progress line after each graph."""
import argparse, copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sgs_gnn_amd as S
from sgs_gnn_amd.stepgraph import StepGraphs

DEV = "cuda:0"
MODE = sys.argv[1] if len(sys.argv) > 1 else "plain"

def mkargs(**kw):
    a = argparse.Namespace(device=DEV, mode="learned", pipeline="hybrid", edge_mlp_type="GCN", conditional=True,
                           sparse_edge_mlp=True, t_init=0.7, t_min=0.5, degree_bias_coef=0.3, reg1=True, reg2=True,
                           regularizer1_coef=1.0, consist_reg_coef=0.5, hybrid_checkpoint=False, drop_rate=0.0, lr=1e-2)
    for k, v in kw.items():
        setattr(a, k, v)
    return a

def setup(p=0.0):
    torch.manual_seed(3); S.fix_seeds(3)
    m = S.GNNModel(24, 32, 5, dropout_prob=p, edge_mlp_type="GCN").to(DEV)
    og = torch.optim.Adam([p_ for n, p_ in m.named_parameters() if "gcn" in n], lr=1e-2)
    oe = torch.optim.Adam([p_ for n, p_ in m.named_parameters() if "edge_prob_mlp" in n], lr=1e-2)
    return m, og, oe

crit = torch.nn.CrossEntropyLoss()
if MODE in ("after_unsampled", "after_unsampled_gc"):
    bs = [S.synthetic_graph(120, E, 24, 5, seed=11 + i, device=DEV) for i, E in enumerate([900, 1500, 700])]
    m2, og2, oe2 = setup()
    for ep in range(4):
        S.train(mkargs(sgs_hipgraph=True), ep, 10, m2, og2, oe2, None, crit, bs, q=5000)
    torch.cuda.synchronize()
    print("phase1 done", flush=True)
    if MODE == "after_unsampled_gc":
        del m2, og2, oe2, bs
        import gc; gc.collect(); torch.cuda.synchronize()
        print("phase1 collected", flush=True)

b = S.synthetic_graph(120, 4000, 24, 5, seed=11, device=DEV)
m, og, oe = setup()
a = mkargs()
sg = StepGraphs.attach(m, "hybrid", a, crit, 800, False)
print("eager", sg.step(b, 0), flush=True)
print("captured", sg.step(b, 0), flush=True)
c = sg.table[next(iter(sg.table))]
held = []
if "keep" in MODE:
    held = [mod._lin_cache for mod in m.modules() if hasattr(mod, "_lin_cache")]
from sgs_gnn_amd.training import sampled_forward, learned_loss, _ce
SYNC = "sync" in MODE
def maybe_sync():
    if SYNC:
        torch.cuda.synchronize()
def eager_work():
    if "eager" not in MODE:
        return
    for p_ in m.parameters():
        p_.grad = None
    st = sampled_forward("hybrid", a, m, b, 800, False)
    loss = learned_loss(a, crit, st, b)
    loss.backward()
    if "opt" in MODE:
        oe.step(); og.step()
for it in range(8):
    maybe_sync(); c.g1.replay(); maybe_sync()
    k = {n: (None if t is None else t.clone()) for n, t in c.keep.items()}
    cnt = c.cbuf.tolist()
    pf = k["edge_probs_full"]
    print(it, "g1 ok", cnt, "p[min,max,nan]", float(pf.min()), float(pf.max()), bool(torch.isnan(pf).any()),
          "eid", int(k["eid"].min()), int(k["eid"].max()), "rsei", int(k["rsei"].min()), int(k["rsei"].max()), flush=True)
    eager_work()
    maybe_sync(); c.g2l.replay(); maybe_sync()
    gl = {i: g.clone() for i, g in c.grads_l.items()}
    eager_work()
    maybe_sync(); c.g2r.replay(); maybe_sync()
    gr = {i: g.clone() for i, g in c.grads_r.items()}
    eager_work()
    torch.cuda.synchronize()
    print(it, "iter ok", float(c.loss_l), float(c.loss_r), flush=True)
print("OK", MODE, flush=True)
