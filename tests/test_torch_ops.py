"""torch.ops.sgs.* (sgs_gnn_amd/torch_ops.py): the hot-path kernels as PyTorch custom operators.
CPU: the operators are registered with schemas and fake kernels (shape inference under FakeTensorMode, no GPU needed).
GPU: torch.library.opcheck (schema, fake kernel, autograd registration) and values / gradients equal to the sgs_gnn_amd.ops path."""
import pytest
import torch

DEV = "cuda:0"


def test_operators_are_registered_with_fake_kernels():
    import sgs_gnn_amd  # noqa: F401
    from torch._subclasses.fake_tensor import FakeTensorMode
    for name in ("sample_topq", "edge_score", "edge_score_backward", "gcn_propagate", "gcn_propagate_backward", "gat_propagate",
                 "gat_propagate_backward"):
        assert hasattr(torch.ops.sgs, name), name
    assert "Tensor node_codes" in str(torch.ops.sgs.edge_score.default._schema)
    with FakeTensorMode():
        N, H, E, q = 10, 8, 33, 5
        codes = torch.empty(N, H, device="cuda")
        ei = torch.empty(2, E, dtype=torch.int64, device="cuda")
        W1, b1, w2, b2 = torch.empty(H, 2 * H, device="cuda"), torch.empty(H, device="cuda"), torch.empty(H, device="cuda"), torch.empty(1, device="cuda")
        p = torch.ops.sgs.edge_score(codes, ei, W1, b1, w2, b2, 0.3, 1, 0, True)
        assert p.shape == (E,) and p.dtype == torch.float32
        mask, eid, sei = torch.ops.sgs.sample_topq(p, None, 0.3, q, True, None, 1, 2, ei)
        assert mask.shape == (E,) and mask.dtype == torch.bool and eid.shape == (q,) and sei.shape == (2, q)
        assert torch.ops.sgs.gcn_propagate(codes, ei, p, b1).shape == (N, H)
        assert torch.ops.sgs.gat_propagate(codes, torch.empty(N, device="cuda"), torch.empty(N, device="cuda"), ei, None, 0.2).shape == (N, H)


@pytest.mark.gpu
def test_custom_ops_match_the_ops_layer_and_pass_opcheck():
    import sgs_gnn_amd as S
    ops = S.ops
    g = torch.Generator().manual_seed(0)
    N, H, E, q = 60, 16, 700, 150
    codes = torch.relu(torch.randn(N, H, generator=g)).to(DEV)
    ei = torch.randint(0, N, (2, E), generator=g).to(DEV)
    W1 = (torch.randn(H, 2 * H, generator=g) * 0.2).to(DEV)
    b1, w2, b2 = (torch.randn(H, generator=g) * 0.1).to(DEV), (torch.randn(H, generator=g) * 0.3).to(DEV), torch.zeros(1, device=DEV)
    # --- edge_score: value and gradients against the autograd.Function path
    la = [t.clone().requires_grad_(True) for t in (codes, W1, b1, w2, b2)]
    lb = [t.clone().requires_grad_(True) for t in (codes, W1, b1, w2, b2)]
    pa = torch.ops.sgs.edge_score(la[0], ei, la[1], la[2], la[3], la[4], 0.3, 5, 0, True)
    pb = ops.edge_score(lb[0], lb[1], lb[2], lb[3].reshape(1, -1), lb[4], ei, p=0.3, seed=5, site=2)
    assert torch.equal(pa, pb)
    gp = torch.randn(E, generator=g).to(DEV)
    pa.backward(gp)
    pb.backward(gp)
    for a, b in zip(la, lb):
        torch.testing.assert_close(a.grad, b.grad, rtol=1e-5, atol=1e-7)
    # eval mode: no dropout
    assert torch.equal(torch.ops.sgs.edge_score(codes, ei, W1, b1, w2, b2, 0.3, 5, 0, False),
                       ops.edge_score(codes, W1, b1, w2.reshape(1, -1), b2, ei))
    # --- sample_topq against the ops layer
    noise = ops.exp_noise(3, 1, E, DEV)
    prior = torch.softmax(torch.rand(E, generator=g), 0).to(DEV)
    mask, eid, sei = torch.ops.sgs.sample_topq(pa.detach(), prior, 0.3, q, False, noise, 0, 0, ei)
    r = ops.sample_topq(ops.SAMPLE_LEARNED, pa.detach(), prior, 0.3, q, ei, noise=noise)
    assert torch.equal(mask, r.mask) and torch.equal(eid, r.eid) and torch.equal(sei, r.edge_index) and torch.equal(sei, ei[:, mask])
    # --- gcn_propagate: values / gradients against ops.gcn_norm + ops.gcn_propagate
    x = torch.randn(N, H, generator=g).to(DEV)
    w = torch.rand(E, generator=g).to(DEV)
    bias = torch.randn(H, generator=g).to(DEV)
    xa, wa, ba = (t.clone().requires_grad_(True) for t in (x, w, bias))
    xb, wb, bb = (t.clone().requires_grad_(True) for t in (x, w, bias))
    ya = torch.ops.sgs.gcn_propagate(xa, ei, wa, ba)
    yb = ops.gcn_propagate(xb, ops.gcn_norm(ops.get_graph(ei, N), wb), bb)
    assert torch.equal(ya, yb)
    gy = torch.randn(N, H, generator=g).to(DEV)
    ya.backward(gy)
    yb.backward(gy)
    for a, b in ((xa, xb), (wa, wb), (ba, bb)):
        torch.testing.assert_close(a.grad, b.grad, rtol=1e-5, atol=1e-6)
    # --- gat_propagate
    a_s, a_d = torch.randn(N, generator=g).to(DEV), torch.randn(N, generator=g).to(DEV)
    t1 = [t.clone().requires_grad_(True) for t in (x, a_s, a_d, bias)]
    t2 = [t.clone().requires_grad_(True) for t in (x, a_s, a_d, bias)]
    y1 = torch.ops.sgs.gat_propagate(t1[0], t1[1], t1[2], ei, t1[3], 0.2)
    y2 = ops.gat_aggregate(t2[0], t2[1], t2[2], t2[3], ops.get_graph(ei, N), 0.2)
    assert torch.equal(y1, y2)
    y1.backward(gy)
    y2.backward(gy)
    for a, b in zip(t1, t2):
        torch.testing.assert_close(a.grad, b.grad, rtol=1e-5, atol=1e-6)
    # --- torch.library.opcheck: schema, fake kernel and autograd registration of every operator
    chk = ("test_schema", "test_faketensor", "test_autograd_registration")
    torch.library.opcheck(torch.ops.sgs.edge_score, (codes.clone().requires_grad_(True), ei, W1, b1, w2, b2, 0.3, 5, 0, True), test_utils=chk)
    torch.library.opcheck(torch.ops.sgs.sample_topq, (pa.detach(), prior, 0.3, q, False, noise, 0, 0, ei), test_utils=chk)
    torch.library.opcheck(torch.ops.sgs.gcn_propagate, (x.clone().requires_grad_(True), ei, w.clone().requires_grad_(True), bias), test_utils=chk)
    torch.library.opcheck(torch.ops.sgs.gat_propagate, (x.clone().requires_grad_(True), a_s, a_d, ei, bias, 0.2), test_utils=chk)
    # --- no CPU kernels: a CPU tensor raises
    with pytest.raises(RuntimeError):
        torch.ops.sgs.edge_score(codes.cpu(), ei.cpu(), W1.cpu(), b1.cpu(), w2.cpu(), b2.cpu(), 0.0, 0, 0, False)
