"""Debug driver: mimic test_one_capture_serves_partitions_of_different_sizes; variants A/B/C/D."""
import argparse, faulthandler, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def run(variant):
    import torch
    faulthandler.enable()
    import sgs_gnn_amd as S
    from sgs_gnn_amd.stepgraph import StepGraphs
    from sgs_gnn_amd.training import _ce
    DEV = "cuda:0"
    a = argparse.Namespace(device=DEV, mode="learned", pipeline="hybrid", edge_mlp_type="GCN", conditional=True, sparse_edge_mlp=True, t_init=0.7,
                           t_min=0.5, degree_bias_coef=0.3, reg1=True, reg2=True, regularizer1_coef=1.0, consist_reg_coef=0.5,
                           hybrid_checkpoint=False, drop_rate=0.0, lr=1e-2)
    torch.manual_seed(3); S.fix_seeds(3)
    m = S.GNNModel(24, 32, 5, dropout_prob=0.0, edge_mlp_type="GCN").to(DEV)
    crit = torch.nn.CrossEntropyLoss()
    shapes = [(150, 6100), (90, 2600), (120, 4000), (110, 900), (140, 1500), (100, 700)]
    bs = [S.synthetic_graph(n, E, 24, 5, seed=40 + i, device=DEV) for i, (n, E) in enumerate(shapes)]
    sg = StepGraphs.attach(m, "hybrid", a, crit, 1000, False, loader=bs)
    sg.debug_keep = True
    params = list(m.parameters())
    for b in bs:
        print("  forward", tuple(b.x.shape), b.edge_index.shape[1], flush=True)
        h = sg.forward(b)
        c = h.c
        if h.sampled:
            cnt = h.gate_counts()
            if "manual" in variant:
                c.g2l.replay(); c.g2r.replay(); sg.host_epoch += 2
            else:
                h.backward(True)
            torch.cuda.synchronize()
            if "eager" in variant:
                ctx = torch.cuda.stream(sg.stream) if "onstream" in variant else torch.cuda.stream(torch.cuda.current_stream())
                with ctx:
                    for p in params:
                        p.grad = None
                    ro = m(b, c.keep["rsei"])
                    _ce(crit, ro, b).backward()
                torch.cuda.synchronize()
        else:
            h.backward(None)
        for p in params:
            p.grad = None
        torch.cuda.synchronize()
        print("    ok", flush=True)
    print("captures", sg.captures, flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        for v in ["manual", "eager", "eager+onstream", "manual+eager", "manual+eager+onstream"]:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), v], capture_output=True, text=True, timeout=300)
            print(f"=== {v}: rc={r.returncode}")
            print(r.stdout[-600:])
            if r.returncode != 0:
                print(r.stderr[-1200:])
