"""Debug driver (not part of the product): which sequence of slot captures crashes?  One variant per process."""
import argparse
import faulthandler
import subprocess
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def run(variant):
    import torch
    faulthandler.enable()
    import sgs_gnn_amd as S
    from sgs_gnn_amd.stepgraph import StepGraphs
    DEV = "cuda:0"
    a = argparse.Namespace(device=DEV, mode="learned", pipeline="hybrid", edge_mlp_type="GCN", conditional=True, sparse_edge_mlp=True, t_init=0.7,
                           t_min=0.5, degree_bias_coef=0.3, reg1=True, reg2=True, regularizer1_coef=1.0, consist_reg_coef=0.5,
                           hybrid_checkpoint=False, drop_rate=0.0, lr=1e-2)
    torch.manual_seed(3)
    S.fix_seeds(3)
    m = S.GNNModel(24, 32, 5, dropout_prob=0.0, edge_mlp_type="GCN").to(DEV)
    crit = torch.nn.CrossEntropyLoss()
    shapes = {"full": [(150, 6100), (90, 2600), (120, 4000), (110, 900), (140, 1500), (100, 700)],
              "sameN": [(150, 6100), (150, 2600), (150, 4000), (150, 900), (150, 1500), (150, 700)],
              "unsampled_only": [(110, 900), (100, 700), (150, 800)],
              "unsampled_first": [(110, 900), (150, 6100), (100, 700), (90, 2600)],
              "one_each": [(150, 6100), (110, 900)],
              "big_unsampled": [(150, 6100), (150, 3000)],
              }[variant.split("+")[0]]
    q = 5000 if variant.startswith("big_unsampled") else 1000
    if variant.startswith("big_unsampled"):
        shapes = [(150, 16100), (150, 3000)]
    bs = [S.synthetic_graph(n, E, 24, 5, seed=40 + i, device=DEV) for i, (n, E) in enumerate(shapes)]
    sg = StepGraphs.attach(m, "hybrid", a, crit, q, False, loader=bs)
    sg.debug_keep = "+keep" in variant
    side = torch.cuda.Stream() if "+side" in variant else None
    for b in bs:
        print("  forward", tuple(b.x.shape), b.edge_index.shape[1], flush=True)
        if side is not None:
            with torch.cuda.stream(side):
                loss, won = sg.step(b, 0)
        else:
            loss, won = sg.step(b, 0)
        torch.cuda.synchronize()
        print("    ok", float(loss), won, flush=True)
    print("captures", sg.captures, flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        for v in ["full+keep", "full", "sameN", "unsampled_only", "unsampled_first", "one_each", "big_unsampled", "one_each+side"]:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), v], capture_output=True, text=True, timeout=300)
            print(f"=== {v}: rc={r.returncode}")
            print(r.stdout[-1500:])
            if r.returncode != 0:
                print(r.stderr[-1800:])
