"""GPU: sgs_gnn_amd.FusedAdam (one launch per parameter group) against torch.optim.Adam."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _params(seed, shapes):
    g = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter(torch.randn(*s, generator=g).to(DEV)) for s in shapes]


@pytest.mark.parametrize("wd,maximize", [(0.0, False), (5e-4, False), (0.0, True)])
def test_fused_adam_matches_torch_adam(wd, maximize):
    import sgs_gnn_amd as S
    shapes = [(256, 602), (256,), (41, 256), (41,), (3, 5, 7), (5000,)]
    pa, pb = _params(0, shapes), _params(0, shapes)
    oa = torch.optim.Adam(pa, lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=wd, maximize=maximize)
    ob = S.FusedAdam(pb, lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=wd, maximize=maximize)
    g = torch.Generator().manual_seed(1)
    for it in range(25):
        # like optimizer_gnn in the reference: some parameters get no gradient on some steps and must be skipped,
        # which makes their own step counter (bias correction) lag behind the others'
        for i, (a, b) in enumerate(zip(pa, pb)):
            if i >= 4 and it % 3 == 1:
                a.grad = b.grad = None
                continue
            gr = torch.randn(a.shape, generator=g).to(DEV) * (1.0 + i)
            a.grad, b.grad = gr.clone(), gr.clone()
        oa.step()
        ob.step()
    torch.cuda.synchronize()
    for a, b in zip(pa, pb):
        torch.testing.assert_close(b, a, rtol=2e-5, atol=2e-7)
    sa, sb = oa.state_dict()["state"], ob.state_dict()["state"]
    for k in sa:
        assert float(sb[k]["step"]) == float(sa[k]["step"])
        torch.testing.assert_close(sb[k]["exp_avg"], sa[k]["exp_avg"], rtol=1e-5, atol=1e-6)      # a few ulp: lerp vs fma forms
        torch.testing.assert_close(sb[k]["exp_avg_sq"], sa[k]["exp_avg_sq"], rtol=1e-5, atol=1e-6)
    # state dicts are interchangeable with torch.optim.Adam(capturable=True)
    oc = torch.optim.Adam(_params(0, shapes), lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=wd, maximize=maximize, capturable=True)
    oc.load_state_dict(ob.state_dict())


def test_fused_adam_more_tensors_than_one_launch_and_capture():
    import sgs_gnn_amd as S
    shapes = [(17, 3)] * 30 + [(4100,)]
    pa, pb = _params(2, shapes), _params(2, shapes)
    oa = torch.optim.Adam(pa, lr=1e-2)
    ob = S.FusedAdam(pb, lr=1e-2)
    grads = [torch.randn_like(p) for p in pa]
    for a, b, gr in zip(pa, pb, grads):
        a.grad, b.grad = gr.clone(), gr.clone()
    oa.step(); ob.step()                              # eager (also initialises the state before the capture)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    gph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gph, stream=s):
        ob.step()
    for _ in range(3):
        oa.step()
        gph.replay()
    torch.cuda.synchronize()
    for a, b in zip(pa, pb):
        torch.testing.assert_close(b, a, rtol=2e-5, atol=2e-7)
    assert float(ob.state[pb[0]]["step"]) == 4.0


def test_fused_adam_rejects_cpu_tensors():
    import sgs_gnn_amd as S
    p = torch.nn.Parameter(torch.zeros(3))
    p.grad = torch.ones(3)
    with pytest.raises(RuntimeError):
        S.FusedAdam([p]).step()


def test_fused_adam_device_gate():
    """gate == 0 leaves the gated tensors and their step counters untouched; other tensors step; gate != 0 steps all."""
    import sgs_gnn_amd as S
    pa, pb = _params(4, [(50, 7), (300,)]), _params(4, [(50, 7), (300,)])
    oa, ob = torch.optim.Adam(pa, lr=1e-2), S.FusedAdam(pb, lr=1e-2)
    gate = torch.zeros(1, device=DEV)
    g = torch.Generator().manual_seed(9)
    for it in range(6):
        grads = [torch.randn(a.shape, generator=g).to(DEV) for a in pa]
        open_ = it % 2 == 0
        gate.fill_(2.0 if open_ else 0.0)
        for a, b, gr in zip(pa, pb, grads):
            b.grad = gr.clone()
        pa[0].grad = grads[0].clone() if open_ else None           # torch: the gated tensor simply has no gradient on closed steps
        pa[1].grad = grads[1].clone()
        oa.step()
        ob.step(gate=gate, gated={pb[0]})
    torch.cuda.synchronize()
    for a, b in zip(pa, pb):
        torch.testing.assert_close(b, a, rtol=2e-5, atol=2e-7)
    assert float(ob.state[pb[0]]["step"]) == 3.0 and float(ob.state[pb[1]]["step"]) == 6.0


def _single_batch_run(S, Adam, steps, hipgraph=False, evaluate_after=False):
    """`cluster_loader = [data]` (main.py:67: graphs below the METIS threshold): ONE resident batch, visited every step."""
    import argparse
    import contextlib
    import io
    b = S.synthetic_graph(300, 9000, 12, 5, seed=3, train_frac=0.5, device=DEV)
    torch.manual_seed(0)
    m = S.GNNModel(12, 32, 5, dropout_prob=0.0, edge_mlp_type="GCN").to(DEV)
    og = Adam([p for n, p in m.named_parameters() if "gcn" in n], lr=1e-2)
    oe = Adam([p for n, p in m.named_parameters() if "edge_prob_mlp" in n], lr=1e-2)
    oa = torch.optim.Adam(m.parameters(), lr=1e-2)
    args = argparse.Namespace(device=DEV, mode="learned", pipeline="hybrid", conditional=True, sparse_edge_mlp=True, t_init=0.7, t_min=0.5,
                              degree_bias_coef=0.3, reg1=True, reg2=True, regularizer1_coef=1.0, consist_reg_coef=0.5,
                              hybrid_checkpoint=False, sgs_hipgraph=hipgraph, num_samples_eval=2)
    S.fix_seeds(7)
    with contextlib.redirect_stdout(io.StringIO()):
        for ep in range(steps):
            S.train(args, ep, steps, m, og, oe, oa, torch.nn.CrossEntropyLoss(), [b], q=2000)
    ev = None
    if evaluate_after:
        S.fix_seeds(99)
        ev = S.ensemble_evaluate(args, m, [b], DEV, q=2000, mode="learned")
    return m, b, args, ev


def test_fused_adam_on_one_resident_batch_is_not_served_stale_memos():
    """GCNConv memoises x W^T within a step, keyed on tensor identity + version; FusedAdam writes W through raw pointers.  With one
    resident batch (same x every step) a memo that survived the optimiser step would make every forward after the first ignore the
    update.  Three eager steps with FusedAdam must track torch.optim.Adam (same noise streams, dropout 0)."""
    import sgs_gnn_amd as S
    ma, _, _, _ = _single_batch_run(S, torch.optim.Adam, 3)
    mb, _, _, _ = _single_batch_run(S, S.FusedAdam, 3)
    for (k, a), (_, b) in zip(ma.state_dict().items(), mb.state_dict().items()):
        torch.testing.assert_close(b, a, rtol=1e-4, atol=2e-6, msg=lambda s: f"{k}: {s}")
    # and the parameters did move on every step: three steps of lr = 1e-2 differ from one
    m1, _, _, _ = _single_batch_run(S, S.FusedAdam, 1)
    assert float((m1.gcn1.lin.weight - mb.gcn1.lin.weight).detach().abs().max()) > 5e-3


def test_evaluate_after_a_replayed_epoch_sees_the_current_weights():
    """After HIP-graph training (the optimiser steps run inside replayed graphs: no version counter moves), evaluate() on the same
    resident batch must use the CURRENT weights: compare with evaluating a fresh model that was loaded with the same state_dict."""
    import sgs_gnn_amd as S
    m, b, args, ev = _single_batch_run(S, S.FusedAdam, 5, hipgraph=True, evaluate_after=True)
    torch.manual_seed(1)
    fresh = S.GNNModel(12, 32, 5, dropout_prob=0.0, edge_mlp_type="GCN").to(DEV)
    fresh.load_state_dict(m.state_dict())
    S.fix_seeds(99)
    b2 = S.Batch(**{k: (v.clone() if torch.is_tensor(v) else v) for k, v in b.__dict__.items() if not k.startswith("_sgs")})
    ev2 = S.ensemble_evaluate(args, fresh, [b2], DEV, q=2000, mode="learned")
    assert ev == ev2


def test_step_many_equals_sequential_steps_of_overlapping_optimisers_and_ticks():
    """FusedAdam.step_many((oe, og), tick=...) == oe.step(); og.step(); loss_tick(...) in ONE launch: the two optimisers of the reference
    overlap (main.py:100-109, 122: 'gcn' also matches edge_prob_mlp.gcn*), the shared tensors step twice per call with the same gradient,
    tensors without a gradient are skipped, different hyper-parameters per optimiser are honoured; also replayed from a HIP graph."""
    import sgs_gnn_amd as S
    shapes = [(40, 9), (300,), (33, 5), (2100,), (7,)]
    pa, pb = _params(5, shapes), _params(5, shapes)
    # optimiser 1 holds tensors 0, 1, 2; optimiser 2 holds 1, 2, 3, 4 (1 and 2 shared)
    oa1 = torch.optim.Adam([pa[0], pa[1], pa[2]], lr=2e-3, betas=(0.9, 0.99), weight_decay=1e-3)
    oa2 = torch.optim.Adam([pa[1], pa[2], pa[3], pa[4]], lr=1e-2)
    ob1 = S.FusedAdam([pb[0], pb[1], pb[2]], lr=2e-3, betas=(0.9, 0.99), weight_decay=1e-3)
    ob2 = S.FusedAdam([pb[1], pb[2], pb[3], pb[4]], lr=1e-2)
    loss_sum = torch.zeros((), device=DEV)
    loss = torch.full((), 0.25, device=DEV)
    epoch = torch.full((1,), 7, dtype=torch.int64, device=DEV)
    g = torch.Generator().manual_seed(3)
    for it in range(4):
        grads = [torch.randn(a.shape, generator=g).to(DEV) for a in pa]
        for i, (a, b) in enumerate(zip(pa, pb)):
            has = not (it == 2 and i == 4)                       # one step where tensor 4 has no gradient
            a.grad = grads[i].clone() if has else None
            b.grad = grads[i].clone() if has else None
        oa1.step(); oa2.step()
        S.FusedAdam.step_many((ob1, ob2), tick=(loss_sum, loss, epoch))
    torch.cuda.synchronize()
    for a, b in zip(pa, pb):
        torch.testing.assert_close(b, a, rtol=2e-5, atol=2e-7)
    assert float(loss_sum) == 1.0 and int(epoch) == 11
    assert float(ob1.state[pb[1]]["step"]) == 4.0 and float(ob2.state[pb[1]]["step"]) == 4.0 and float(ob2.state[pb[4]]["step"]) == 3.0
    # captured: the same launch replayed
    for a, b in zip(pa, pb):
        a.grad = torch.ones_like(a)
        b.grad = torch.ones_like(b)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    gph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gph, stream=s):
        S.FusedAdam.step_many((ob1, ob2), tick=(loss_sum, loss, epoch))
    for _ in range(3):
        oa1.step(); oa2.step()
        gph.replay()
    torch.cuda.synchronize()
    for a, b in zip(pa, pb):
        torch.testing.assert_close(b, a, rtol=2e-5, atol=2e-7)
    assert float(loss_sum) == 1.75 and int(epoch) == 14
