"""One tiny invocation of the hot path on cuda:0 checked against the oracle (used by
__graft_entry__.smoke): a full hybrid training step (prior draw -> EdgeProbGCN scores -> learned
draw -> weighted GCN -> gate -> losses -> backward -> Adam) replayed on a reference fixture."""
import argparse
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import sgs_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def run_smoke(pkg):
    dev = "cuda:0"
    ops = pkg.ops
    # (1) sampler vs oracle on random data
    g = torch.Generator().manual_seed(0)
    E, q = 5000, 1000
    p = torch.sigmoid(torch.randn(E, generator=g))
    prior = F.softmax(torch.rand(E, generator=g), dim=0)
    noise = torch.empty(E).exponential_(1, generator=g)
    ei = torch.randint(0, 300, (2, E), generator=g)
    r = ops.sample_topq(ops.SAMPLE_LEARNED, p.to(dev), prior.to(dev), 0.3, q, ei.to(dev), noise=noise.to(dev))
    mask, _ = O.gumbel_softmax_sampling(prior, p, q, 0.3, False, noise, Z=r.stats[0].cpu())
    assert torch.equal(r.mask.cpu(), mask), "sampler mask differs from oracle"
    assert torch.equal(r.edge_index.cpu(), ei[:, mask])

    # (2) one full hybrid step vs the oracle, on the inputs/noise of a reference fixture
    fx = torch.load(os.path.join(GOLDEN, "pipeline_hybrid_gcn.pt"), weights_only=True)
    st = fx["steps"][0]
    m = pkg.GNNModel(fx["x"].shape[1], 16, 5, dropout_prob=0.0, edge_mlp_type="GCN")
    m.load_state_dict(fx["state0"])
    m = m.to(dev)
    opt_gnn = torch.optim.Adam([p_ for n, p_ in m.named_parameters() if "gcn" in n], lr=1e-3)
    opt_edge = torch.optim.Adam([p_ for n, p_ in m.named_parameters() if "edge_prob_mlp" in n], lr=1e-3)
    opt_all = torch.optim.Adam(m.parameters(), lr=1e-3)
    b = pkg.Batch(x=fx["x"], edge_index=fx["edge_index"], y=fx["y"], train_mask=fx["train_mask"], prob=fx["prob"]).to(dev)
    args = argparse.Namespace(device=dev, mode="learned", pipeline="hybrid", conditional=True, sparse_edge_mlp=False, t_init=0.7,
                              t_min=0.5, degree_bias_coef=0.3, reg1=True, reg2=True, regularizer1_coef=1.0, consist_reg_coef=0.5,
                              hybrid_checkpoint=False)
    args._sgs_noise = {"prior": st["noise"][0].to(dev), "sample": st["noise"][1].to(dev)}
    args._sgs_trace = tr = {}
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        ret = pkg.train(args, 0, 10, m, opt_gnn, opt_edge, opt_all, nn.CrossEntropyLoss(), [b], q=fx["q"])

    P = {k: v.clone().requires_grad_(True) for k, v in fx["state0"].items()}
    cfg = O.StepConfig(pipeline="hybrid", scorer="GCN", q=fx["q"], conditional=True)
    nz = O.StepNoise(prior_noise=st["noise"][0], sample_noise=st["noise"][1])
    R = O.learned_step_forward(P, dict(x=fx["x"], edge_index=fx["edge_index"], y=fx["y"], train_mask=fx["train_mask"],
                                       prob=fx["prob"]), cfg, nz)
    assert torch.equal(tr["sample"].mask.cpu(), R["mask"]), "learned draw differs from oracle"
    assert float((tr["learned_out"].cpu() - R["learned_out"].detach()).abs().max()) < 1e-4, "logits differ"
    assert abs(ret[0] - float(R["loss"].detach())) < 1e-4, "loss differs"
    assert int(tr["update_edge_mlp"]) == int(R["update_edge_mlp"])
