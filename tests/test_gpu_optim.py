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
