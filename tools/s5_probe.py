"""Config 5 at full Reddit scale on ONE MI355X (N = 232 965, F = 602, C = 41, E = 114.6 M candidate edges, q = 22.9 M): one eager
hybrid step of train() and one train_step_sharded() (world size 1), wall-clock per step and peak HBM.  Run under
`rocprofv3 --kernel-trace --stats` for the per-kernel figures (profiles/r02_s5_*)."""
import argparse, contextlib, io, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sgs_gnn_amd as S

dev = "cuda:0"
N, F_, C, H = 232_965, 602, 41, 256
E_target = int(os.environ.get("S5_EDGES", "114615892"))
t0 = time.perf_counter()
b = S.synthetic_graph(N, E_target, F_, C, seed=77, train_frac=0.66, power=float(os.environ.get("S5_POWER", "0.35")), device=dev)
torch.cuda.synchronize()
E = b.edge_index.shape[1]
q = int(E * 0.2)
print(json.dumps({"graph_build_s": round(time.perf_counter() - t0, 2), "E": E, "q": q, "hbm_GiB": round(torch.cuda.memory_allocated() / 2**30, 2)}), flush=True)
torch.manual_seed(0)
m = S.GNNModel(F_, H, C, dropout_prob=0.3, edge_mlp_type="GCN").to(dev)
og = S.FusedAdam([p for n, p in m.named_parameters() if "gcn" in n], lr=1e-3)
oe = S.FusedAdam([p for n, p in m.named_parameters() if "edge_prob_mlp" in n], lr=1e-3)
args = argparse.Namespace(device=dev, mode="learned", pipeline="hybrid", conditional=True, sparse_edge_mlp=True, t_init=0.7, t_min=0.5,
                          degree_bias_coef=0.3, reg1=True, reg2=True, regularizer1_coef=1.0, consist_reg_coef=0.5, hybrid_checkpoint=True)
crit = torch.nn.CrossEntropyLoss()
S.fix_seeds(1)
times = []
for ep in range(int(os.environ.get("S5_STEPS", "3"))):
    torch.cuda.synchronize(); t = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        r = S.train(args, ep, 10, m, og, oe, None, crit, [b], q=q)
    torch.cuda.synchronize(); times.append(round(time.perf_counter() - t, 4))
    print(json.dumps({"train_step_s": times[-1], "ret": [round(r[0], 4), r[2], r[3]], "peak_GiB": round(torch.cuda.max_memory_allocated() / 2**30, 1)}), flush=True)
if os.environ.get("S5_SHARDED", "1") == "1":
    from sgs_gnn_amd import sharded as sh
    shard = sh.EdgeShard(b, 0, 1)
    for ep in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        tr = sh.train_step_sharded(args, m, shard, og, oe, crit, q)
        torch.cuda.synchronize()
        print(json.dumps({"sharded_step_s": round(time.perf_counter() - t, 4), "loss": round(float(tr["loss"]), 4)}), flush=True)
print(json.dumps({"steps_s": times, "sampled_edges_per_s_last": round(q / times[-1], 1), "peak_GiB": round(torch.cuda.max_memory_allocated() / 2**30, 1)}))
