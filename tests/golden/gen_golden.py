#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE in the build
container (`/root/reference`, read-only).  The reference never travels to the GPU box,
so only the resulting vectors (inputs + expected outputs, plain tensors) are committed.

    python tests/golden/gen_golden.py            # rewrites tests/golden/*.pt

What is executed from the reference, unmodified:
  * sampling.py  (gumbel_softmax_sampling, random_edge_sampling)  -- pure torch.
  * model.py, utils.py, training_hybrid.py, training_straight_through.py,
    training_two_pass.py, training.py -- these import `torch_geometric`, which is not
    installed here and cannot be fetched.  The import is satisfied with a minimal module
    object whose ONLY functional member is `GCNConv`, implemented by this repo's oracle
    restatement of the PyG 2.3.1 layer (oracle/sgs_oracle.py: gcn_conv).  Consequently
    these fixtures pin the reference's OWN code (scorer, sampler, pipelines, gate, losses,
    optimiser overlap) but NOT the third-party GCN layer, whose parity stays "unpinned"
    (DESIGN.md).

Randomness: the reference draws from torch's global CPU generator (multinomial ->
exponential_, nn.Dropout -> bernoulli_).  We wrap `torch.multinomial` and `F.dropout` so
that the SAME call still runs, and the noise / keep-mask it consumed is additionally
recorded (replayed from the saved generator state and asserted identical), so the oracle
and the HIP path can be fed the very same bits.
"""
import os
import sys
import types
import argparse

import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import sgs_oracle as O  # noqa: E402


# ------------------------------------------------------------------ recorders
class Recorder:
    def __init__(self):
        self.noise = []      # list of (s, q, noise, idx)
        self.drop = []       # list of (shape, p, keep)
        self._real_multinomial = torch.multinomial
        self._real_dropout = F.dropout

    def install(self):
        rec = self

        def multinomial(s, q, replacement=False, **kw):
            assert replacement is False
            st = torch.get_rng_state()
            idx = rec._real_multinomial(s, q, replacement=False, **kw)
            st_after = torch.get_rng_state()
            torch.set_rng_state(st)
            noise = torch.empty_like(s).exponential_(1)
            assert torch.equal(torch.get_rng_state(), st_after)
            chk = torch.topk(s / noise, q).indices
            assert torch.equal(chk, idx), "multinomial != topk(s/Exp(1))"
            rec.noise.append((s.detach().clone(), q, noise, idx.clone()))
            return idx

        def dropout(x, p=0.5, training=True, inplace=False):
            if not training or p == 0.0:
                return rec._real_dropout(x, p, training, inplace)
            st = torch.get_rng_state()
            out = rec._real_dropout(x, p, training, False)
            st_after = torch.get_rng_state()
            torch.set_rng_state(st)
            keep = rec._real_dropout(torch.ones_like(x), p, True, False) != 0
            assert torch.equal(torch.get_rng_state(), st_after)
            assert torch.allclose(out, x * keep / (1 - p))
            rec.drop.append((tuple(x.shape), p, keep))
            return out

        torch.multinomial = multinomial
        F.dropout = dropout
        torch.nn.functional.dropout = dropout

    def uninstall(self):
        torch.multinomial = self._real_multinomial
        F.dropout = self._real_dropout
        torch.nn.functional.dropout = self._real_dropout

    def clear(self):
        self.noise, self.drop = [], []


# ------------------------------------------------------------------ torch_geometric stand-in
class GCNConv(nn.Module):
    """State-dict compatible with PyG's GCNConv (`lin.weight`, `bias`); forward =
    oracle restatement.  Third-party layer: parity unpinned."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.lin = nn.Linear(in_channels, out_channels, bias=False)
        self.bias = nn.Parameter(torch.zeros(out_channels))
        a = (6.0 / (in_channels + out_channels)) ** 0.5
        nn.init.uniform_(self.lin.weight, -a, a)

    def forward(self, x, edge_index, edge_weight=None):
        return O.gcn_conv(x, edge_index, edge_weight, self.lin.weight, self.bias)


def _absent(name):
    class _Absent(nn.Module):
        def __init__(self, *a, **k):
            raise NotImplementedError(f"torch_geometric.{name} is not available in the build container")
    _Absent.__name__ = name
    return _Absent


def install_pyg_stub():
    tg = types.ModuleType("torch_geometric")
    tgnn = types.ModuleType("torch_geometric.nn")
    tgu = types.ModuleType("torch_geometric.utils")
    tgd = types.ModuleType("torch_geometric.data")
    tgnn.GCNConv = GCNConv
    for n in ("GATConv", "GINConv", "SAGEConv", "ChebConv", "GAT", "GIN"):
        setattr(tgnn, n, _absent(n))
    tgu.to_networkx = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError())
    tgd.Data = type("Data", (), {})
    tg.nn, tg.utils, tg.data = tgnn, tgu, tgd
    sys.modules.update({"torch_geometric": tg, "torch_geometric.nn": tgnn,
                        "torch_geometric.utils": tgu, "torch_geometric.data": tgd})


# ------------------------------------------------------------------ synthetic batch
class Batch:
    def __init__(self, **kw):
        self.__dict__.update(kw)

    def to(self, device):
        return self


def make_graph(n, avg_deg, nfeat, ncls, seed, train_frac=0.5):
    g = torch.Generator().manual_seed(seed)
    m = n * avg_deg // 2
    a = torch.randint(0, n, (m,), generator=g)
    b = torch.randint(0, n, (m,), generator=g)
    keep = a != b
    a, b = a[keep], b[keep]
    ei = torch.cat([torch.stack([a, b]), torch.stack([b, a])], dim=1)
    key = torch.unique(ei[0] * n + ei[1])               # coalesced + row-sorted
    ei = torch.stack([key // n, key % n])
    x = torch.randn(n, nfeat, generator=g)
    y = torch.randint(0, ncls, (n,), generator=g)
    # make labels correlate with features a little so the gate is not a coin flip
    x[torch.arange(n), y % nfeat] += 2.0
    tm = torch.rand(n, generator=g) < train_frac
    prob = O.add_degree_prior(ei, n)
    return Batch(x=x, edge_index=ei, y=y, train_mask=tm, val_mask=~tm, test_mask=~tm, prob=prob)


def ref_args(pipeline, scorer, drop, conditional):
    return argparse.Namespace(
        device="cpu", mode="learned", pipeline=pipeline, edge_mlp_type=scorer, conditional=conditional,
        sparse_edge_mlp=False, t_init=0.7, t_min=0.5, degree_bias_coef=0.3, reg1=True, reg2=True,
        regularizer1_coef=1.0, consist_reg_coef=0.5, hybrid_checkpoint=False, drop_rate=drop, lr=1e-3)


def sd_clone(model):
    return {k: v.detach().clone() for k, v in model.state_dict().items()}


# ------------------------------------------------------------------ sampler-only fixtures
def gen_sampler(rec):
    import sampling as ref_sampling
    cases = []
    for (E, q, istest, seed) in [(1000, 200, False, 1), (1000, 200, True, 2), (37, 36, False, 3),
                                 (4096, 1, False, 4), (513, 512, True, 5), (8000, 1600, False, 6)]:
        torch.manual_seed(seed)
        p = torch.sigmoid(torch.randn(E))
        prior = F.softmax(torch.rand(E) * 3, dim=0)
        ei = torch.randint(0, 64, (2, E))
        rec.clear()
        mask, w = ref_sampling.gumbel_softmax_sampling(Batch(prob=prior), p, ei, q=q, degree_bias_coef=0.3,
                                                       istest=istest)
        (s, _, noise, idx), = rec.noise
        cases.append(dict(E=E, q=q, istest=istest, p=p, prior=prior, edge_index=ei, noise=noise,
                          samples=s, Z=p.sum(), idx=idx, mask=mask, w=w,
                          sampled_edge_index=ei[:, mask]))
    # tie / degenerate cases where the rule "lowest edge id wins" matters are oracle-defined,
    # not reference-defined (torch.topk leaves ties unspecified) -> not generated here.
    prior_cases = []
    for (E, q, seed) in [(1000, 200, 11), (5000, 1000, 12)]:
        torch.manual_seed(seed)
        ei = make_graph(200, E // 200, 4, 3, seed).edge_index
        prob = O.add_degree_prior(ei, 200)
        rec.clear()
        rs = F.softmax(prob, dim=-1)
        idx = torch.multinomial(rs, q, replacement=False)      # training_hybrid.py:46-47
        (s, _, noise, _), = rec.noise
        prior_cases.append(dict(E=ei.shape[1], q=q, prob=prob, softmax=rs, noise=noise, idx=idx, edge_index=ei,
                                rsei=ei[:, idx]))
    rperm_cases = []
    torch.manual_seed(21)
    ei = torch.randint(0, 50, (2, 300))
    st = torch.get_rng_state()
    out = ref_sampling.random_edge_sampling(ei, 60)
    torch.set_rng_state(st)
    perm = torch.randperm(300)
    rperm_cases.append(dict(edge_index=ei, q=60, perm=perm, out=out))
    torch.save(dict(learned=cases, prior=prior_cases, randperm=rperm_cases), os.path.join(HERE, "sampler.pt"))
    print("sampler.pt:", len(cases), "learned,", len(prior_cases), "prior,", len(rperm_cases), "randperm cases")


# ------------------------------------------------------------------ pipeline fixtures
def gen_pipeline(rec, name, pipeline, scorer, drop, conditional, nsteps, seed):
    import model as ref_model
    import training as ref_training
    torch.manual_seed(seed)
    b = make_graph(48, 10, 12, 5, seed)
    E = b.edge_index.shape[1]
    q = int(E * 0.2)
    args = ref_args(pipeline, scorer, drop, conditional)
    m = ref_model.GNNModel(12, 16, 5, dropout_prob=drop, edge_mlp_type=scorer)
    # non-zero GCN biases so that parity exercises them
    with torch.no_grad():
        for k, v in m.named_parameters():
            if k.endswith("bias") and "gcn" in k:
                v.uniform_(-0.05, 0.05)
    opt_gnn = torch.optim.Adam([p for n, p in m.named_parameters() if "gcn" in n], lr=args.lr)      # main.py:100
    opt_edge = torch.optim.Adam([p for n, p in m.named_parameters() if "edge_prob_mlp" in n], lr=args.lr)  # :122
    opt_all = torch.optim.Adam(m.parameters(), lr=args.lr, weight_decay=5e-4)                        # :123
    crit = nn.CrossEntropyLoss()

    cap = {}
    h1 = m.edge_prob_mlp.register_forward_hook(lambda mod, i, o: cap.setdefault("scorer_out", []).append(o.detach().clone()))
    h2 = m.register_forward_hook(lambda mod, i, o: cap.setdefault("gnn_out", []).append(
        (i[1].clone(), None if len(i) < 3 or i[2] is None else i[2].detach().clone(), o.detach().clone())))
    tmod = {"hybrid": "training_hybrid", "straight_through": "training_straight_through",
            "two_pass": "training_two_pass"}[pipeline]
    tm = sys.modules[tmod]
    real_sampler = tm.gumbel_softmax_sampling

    def sampler_spy(*a, **k):
        mask, w = real_sampler(*a, **k)
        cap.setdefault("sampler", []).append((mask.clone(), w.detach().clone()))
        return mask, w
    tm.gumbel_softmax_sampling = sampler_spy

    steps = []
    sd0 = sd_clone(m)
    try:
        for epoch in range(nsteps):
            rec.clear()
            cap.clear()
            ret = ref_training.train(args, epoch, 10, m, opt_gnn, opt_edge, opt_all, crit, [b], q=q,
                                     alternate_frequency=0)
            st = dict(ret_loss=float(ret[0]), ret_temperature=float(ret[1]), ret_cond=int(ret[2]), ret_total=int(ret[3]))
            st["noise"] = [n for (_, _, n, _) in rec.noise]
            st["noise_idx"] = [i for (_, _, _, i) in rec.noise]
            st["drop_keep"] = [k for (_, _, k) in rec.drop]
            st["drop_shapes"] = [list(s) for (s, _, _) in rec.drop]
            st["scorer_out"] = cap.get("scorer_out", [])
            st["gnn_edge_index"] = [g[0] for g in cap.get("gnn_out", [])]
            st["gnn_edge_weight"] = [g[1] if g[1] is not None else torch.zeros(0) for g in cap.get("gnn_out", [])]
            st["gnn_out"] = [g[2] for g in cap.get("gnn_out", [])]
            st["mask"] = cap["sampler"][0][0]
            st["st_w"] = cap["sampler"][0][1]
            st["grads"] = {k: (v.grad.detach().clone() if v.grad is not None else torch.zeros(0))
                           for k, v in m.named_parameters()}
            st["state_after"] = sd_clone(m)
            steps.append(st)
    finally:
        tm.gumbel_softmax_sampling = real_sampler
        h1.remove()
        h2.remove()
    fx = dict(name=name, pipeline=pipeline, scorer=scorer, drop=drop, conditional=conditional, q=q,
              x=b.x, edge_index=b.edge_index, y=b.y, train_mask=b.train_mask, prob=b.prob,
              state0=sd0, steps=steps)
    torch.save(fx, os.path.join(HERE, f"pipeline_{name}.pt"))
    print(f"pipeline_{name}.pt: E={E} q={q} steps={nsteps} loss0={steps[0]['ret_loss']:.6f} "
          f"cond={[s['ret_cond'] for s in steps]}")


def gen_pipeline_fullsize(rec, name, pipeline, salt, n=1013, nfeat=602, ncls=41, hid=256, e_target=210_000, q=100_000, nsteps=1,
                          conditional=True):
    """One Reddit-partition-sized step of the REFERENCE's trainer (S3 shapes: n=1013, F=602, H=256, C=41, q=100 000), so that the
    kernels the HIP path auto-selects at production size (bf16x6 scorer forward / backward core, dv.W1a row GEMM, tall-K weight
    gradient GEMM) are compared with the reference itself.  Inputs and the initial state are rebuilt from integer hashes
    (tests/golden/portable.py) on the test side; stored here: the Exp(1) noise both multinomials consumed, and the reference's
    outputs (scores of all E edges, packed masks, logits, loss, gate, all gradients)."""
    import model as ref_model
    import training as ref_training
    import portable as PT
    part = PT.make_partition(n, nfeat, ncls, e_target, salt)
    ei = part["edge_index"]
    E = ei.shape[1]
    prob = PT.degree_prior(ei, n)
    b = Batch(x=part["x"], edge_index=ei, y=part["y"], train_mask=part["train_mask"], val_mask=~part["train_mask"],
              test_mask=~part["train_mask"], prob=prob)
    args = ref_args(pipeline, "GCN", 0.0, conditional)
    torch.manual_seed(salt)
    m = ref_model.GNNModel(nfeat, hid, ncls, dropout_prob=0.0, edge_mlp_type="GCN")
    sd0 = PT.init_state({k: tuple(v.shape) for k, v in m.state_dict().items()}, salt + 5000)
    m.load_state_dict(sd0)
    opt_gnn = torch.optim.Adam([p for n_, p in m.named_parameters() if "gcn" in n_], lr=args.lr)
    opt_edge = torch.optim.Adam([p for n_, p in m.named_parameters() if "edge_prob_mlp" in n_], lr=args.lr)
    opt_all = torch.optim.Adam(m.parameters(), lr=args.lr, weight_decay=5e-4)
    crit = nn.CrossEntropyLoss()
    cap = {}
    h1 = m.edge_prob_mlp.register_forward_hook(lambda mod, i, o: cap.setdefault("scorer_out", []).append(o.detach().clone()))
    h2 = m.register_forward_hook(lambda mod, i, o: cap.setdefault("gnn_out", []).append(
        (i[1].clone(), None if len(i) < 3 or i[2] is None else i[2].detach().clone(), o.detach().clone())))
    tmod = {"hybrid": "training_hybrid", "straight_through": "training_straight_through", "two_pass": "training_two_pass"}[pipeline]
    tm = sys.modules[tmod]
    real_sampler = tm.gumbel_softmax_sampling

    def sampler_spy(*a, **k):
        mask, w = real_sampler(*a, **k)
        cap.setdefault("sampler", []).append((mask.clone(), w.detach().clone()))
        return mask, w
    tm.gumbel_softmax_sampling = sampler_spy
    steps = []
    try:
        for epoch in range(nsteps):
            rec.clear()
            cap.clear()
            ret = ref_training.train(args, epoch, 10, m, opt_gnn, opt_edge, opt_all, crit, [b], q=q, alternate_frequency=0)
            st = dict(ret_loss=float(ret[0]), ret_temperature=float(ret[1]), ret_cond=int(ret[2]), ret_total=int(ret[3]))
            st["noise"] = [n_ for (_, _, n_, _) in rec.noise]
            rmask = torch.zeros(E, dtype=torch.bool)
            if conditional:
                rmask[rec.noise[0][3]] = True
            st["prior_mask_packed"] = PT.pack_mask(rmask)
            st["scorer_out"] = cap["scorer_out"][0].squeeze().clone()
            st["mask_packed"] = PT.pack_mask(cap["sampler"][0][0])
            st["gnn_out"] = [g[2] for g in cap.get("gnn_out", [])]
            st["w_sampled"] = cap["gnn_out"][0][1]
            st["correct"] = [int((g[2].argmax(1) == b.y)[b.train_mask].sum()) for g in cap.get("gnn_out", [])]    # the gate's F1 numerators
            st["grads"] = {k: (v.grad.detach().clone() if v.grad is not None else torch.zeros(0)) for k, v in m.named_parameters()}
            steps.append(st)
    finally:
        tm.gumbel_softmax_sampling = real_sampler
        h1.remove()
        h2.remove()
    fx = dict(name=name, pipeline=pipeline, scorer="GCN", drop=0.0, conditional=conditional, q=q, salt=salt, n=n, nfeat=nfeat, ncls=ncls,
              hid=hid, e_target=e_target, E=E, x_checksum=float(part["x"].double().sum()), ei_checksum=int(ei.sum()),
              state0_checksum=float(sum(v.double().abs().sum() for v in sd0.values())), prob_checksum=float(prob.double().sum()),
              steps=steps)
    torch.save(fx, os.path.join(HERE, f"pipeline_{name}.pt"))
    print(f"pipeline_{name}.pt: E={E} q={q} steps={nsteps} loss0={steps[0]['ret_loss']:.6f} cond={[s_['ret_cond'] for s_ in steps]}")


def main():
    install_pyg_stub()
    sys.path.insert(0, REF)
    rec = Recorder()
    rec.install()
    try:
        gen_sampler(rec)
        for name, pipeline, scorer, drop, cond, nsteps, seed in [
            ("hybrid_gcn", "hybrid", "GCN", 0.0, True, 3, 122),
            ("st_gcn", "straight_through", "GCN", 0.0, True, 3, 102),
            ("twopass_gcn", "two_pass", "GCN", 0.0, True, 3, 103),
            ("hybrid_mlp", "hybrid", "MLP", 0.0, False, 2, 104),
            ("hybrid_gcn_drop", "hybrid", "GCN", 0.3, True, 2, 105),
            ("twopass_mlp", "two_pass", "MLP", 0.0, False, 2, 106),
            ("hybrid_mlp_drop", "hybrid", "MLP", 0.3, False, 2, 107),
        ]:
            gen_pipeline(rec, name, pipeline, scorer, drop, cond, nsteps, seed)
        sys.path.insert(0, HERE)
        gen_pipeline_fullsize(rec, "hybrid_gcn_s3size", "hybrid", salt=33)   # salt chosen so that the gate takes the learned branch
    finally:
        rec.uninstall()


if __name__ == "__main__":
    main()
