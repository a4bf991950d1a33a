"""GPU parity: graph build, gcn_norm, CSR SpMM / SDDMM and their autograd glue vs the oracle."""
import pytest
import torch
import torch.nn.functional as F

from oracle import sgs_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def pkg():
    import sgs_gnn_amd
    return sgs_gnn_amd


def rand_graph(N, E, seed, loops=True):
    g = torch.Generator().manual_seed(seed)
    ei = torch.randint(0, N, (2, E), generator=g)
    if loops and E > 12:
        ei[:, 3] = ei[0, 3]
        ei[:, 7] = ei[0, 3]      # a second self loop on the same node: the last one wins
        ei[:, 11] = ei[0, 11]
    return ei, g


@pytest.mark.parametrize("N,E", [(1, 0), (5, 3), (50, 700), (1013, 100000), (3, 30000), (40000, 90000)])
def test_graph_build_matches_stable_sort(pkg, N, E):
    ei, _ = rand_graph(N, E, N + E)
    gr = pkg.ops.Graph(ei.to(DEV), N)
    torch.cuda.synchronize()
    src, dst = ei[0], ei[1]
    for ptr, col, eid, key, other in [(gr.in_ptr, gr.in_src, gr.in_eid, dst, src), (gr.out_ptr, gr.out_dst, gr.out_eid, src, dst)]:
        want_ptr = torch.zeros(N + 1, dtype=torch.int64)
        want_ptr[1:] = torch.cumsum(torch.bincount(key, minlength=N), 0)
        assert torch.equal(ptr.cpu().long(), want_ptr)
        order = torch.sort(key, stable=True).indices
        assert torch.equal(eid.cpu().long()[:E], order)
        assert torch.equal(col.cpu().long()[:E], other[order])
    want_loop = torch.full((N,), -1, dtype=torch.int64)
    for e in range(E):
        if src[e] == dst[e]:
            want_loop[src[e]] = e
    assert torch.equal(gr.loop_eid.cpu().long()[:N], want_loop)


@pytest.mark.parametrize("N,E,Fin,D", [(40, 300, 9, 16), (40, 300, 9, 41), (200, 5000, 33, 256), (64, 900, 5, 7), (30, 200, 4, 70),
                                       (17, 60, 3, 512), (25, 0, 3, 8)])
@pytest.mark.parametrize("weighted", [True, False])
def test_gcn_conv_forward_backward_vs_oracle(pkg, N, E, Fin, D, weighted):
    from sgs_gnn_amd.model import GCNConv
    ei, g = rand_graph(N, E, N * 3 + D)
    x = torch.randn(N, Fin, generator=g)
    w = torch.rand(E, generator=g) if weighted else None
    conv = GCNConv(Fin, D)
    with torch.no_grad():
        conv.bias.uniform_(-0.5, 0.5)
    W, b = conv.lin.weight.detach().clone(), conv.bias.detach().clone()
    gy = torch.randn(N, D, generator=g)

    # oracle (fp32 and fp64)
    res = {}
    for dt in (torch.float32, torch.float64):
        xo = x.clone().to(dt).requires_grad_(True)
        Wo, bo = W.clone().to(dt).requires_grad_(True), b.clone().to(dt).requires_grad_(True)
        wo = w.clone().to(dt).requires_grad_(True) if weighted else None
        yo = O.gcn_conv(xo, ei, wo, Wo, bo)
        yo.backward(gy.to(dt))
        res[dt] = (yo.detach(), xo.grad, Wo.grad, bo.grad, wo.grad if weighted else None)

    conv = conv.to(DEV)
    xd = x.clone().to(DEV).requires_grad_(True)
    wd = w.clone().to(DEV).requires_grad_(True) if weighted else None
    yd = conv(xd, ei.to(DEV), wd)
    yd.backward(gy.to(DEV))
    got = (yd.detach().cpu(), xd.grad.cpu(), conv.lin.weight.grad.cpu(), conv.bias.grad.cpu(), wd.grad.cpu() if weighted else None)
    for name, a, r32, r64 in zip(["y", "dx", "dW", "db", "dw"], got, res[torch.float32], res[torch.float64]):
        if a is None or a.numel() == 0:
            continue
        # error vs fp64 truth no worse than a few times the fp32 oracle's own error, and <= 1e-4
        scale = float(r64.abs().max()) + 1e-12
        err = float((a.double() - r64).abs().max()) / scale
        err32 = float((r32.double() - r64).abs().max()) / scale
        assert err <= max(5 * err32, 2e-6) and err < 1e-4, f"{name}: rel err {err:.2e} (fp32 oracle {err32:.2e})"


@pytest.mark.parametrize("p", [0.0, 0.3])
def test_gnn_model_two_layers_vs_oracle(pkg, p):
    """GNNModel.forward (model.py:155-164): shared norm across layers, fused ReLU+dropout, edge-weight
    gradient through both layers and through the degree normalisation."""
    from sgs_gnn_amd import model as M
    N, E, Fin, H, C = 120, 2000, 20, 32, 6
    ei, g = rand_graph(N, E, 77)
    x = torch.randn(N, Fin, generator=g)
    w = torch.rand(E, generator=g)
    P = O.init_params(Fin, H, C, "GCN", seed=5)

    class D_:
        pass
    data = D_()
    data.x = x.to(DEV)
    m = M.GNNModel.__new__(M.GNNModel)
    torch.nn.Module.__init__(m)
    m.gcn1, m.gcn2, m.dropout = M.GCNConv(Fin, H), M.GCNConv(H, C), torch.nn.Dropout(p)
    with torch.no_grad():
        m.gcn1.lin.weight.copy_(P["gcn1.lin.weight"]); m.gcn1.bias.copy_(P["gcn1.bias"])
        m.gcn2.lin.weight.copy_(P["gcn2.lin.weight"]); m.gcn2.bias.copy_(P["gcn2.bias"])
    m = m.to(DEV).train()
    M.set_dropout_seed(99)
    seed = (M._DropoutClock.base * 0x9E3779B97F4A7C15 + 1 * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
    keep = pkg.ops.dropout_keep(seed, M.SITE_GNN, N, H, p, DEV).cpu() if p > 0 else None
    if p > 0:
        assert abs(float(keep.float().mean()) - (1 - p)) < 0.03
    wd = w.to(DEV).requires_grad_(True)
    out = m(data, ei.to(DEV), wd)
    gy = torch.randn(N, C, generator=g)
    out.backward(gy.to(DEV))

    Po = {k: v.clone().double().requires_grad_(True) for k, v in P.items()}
    wo = w.double().requires_grad_(True)
    oo = O.gnn_forward(Po, x.double(), ei, wo, p, keep)
    oo.backward(gy.double())
    def rel(a, b):
        return float((a.double().cpu() - b).abs().max()) / (float(b.abs().max()) + 1e-12)
    assert rel(out.detach(), oo.detach()) < 1e-5
    assert rel(wd.grad, wo.grad) < 1e-4
    assert rel(m.gcn1.lin.weight.grad, Po["gcn1.lin.weight"].grad) < 1e-4
    assert rel(m.gcn1.bias.grad, Po["gcn1.bias"].grad) < 1e-4
    assert rel(m.gcn2.lin.weight.grad, Po["gcn2.lin.weight"].grad) < 1e-4
    assert rel(m.gcn2.bias.grad, Po["gcn2.bias"].grad) < 1e-4


def test_propagate_is_run_to_run_deterministic(pkg):
    N, E, D = 1013, 100000, 256
    ei, g = rand_graph(N, E, 5)
    X = torch.randn(N, D, generator=g).to(DEV)
    w = torch.rand(E, generator=g).to(DEV)
    outs = []
    for _ in range(3):
        gr = pkg.ops.Graph(ei.to(DEV), N)
        nm = pkg.ops.gcn_norm(gr, w)
        outs.append(pkg.ops.gcn_propagate(X, nm))
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("K,M,N", [(1013, 256, 602), (1013, 41, 256), (7, 5, 3), (2000, 256, 256), (333, 70, 8710), (64, 32, 32), (1, 4, 4),
                                   (20011, 256, 256), (9001, 132, 70), (8192, 4, 2)])
def test_gemm_tn_weight_gradient(pkg, K, M, N):
    """dW = dY^T X on the f32 matrix cores vs fp64."""
    g = torch.Generator().manual_seed(K + M + N)
    dY = torch.randn(K, M, generator=g)
    X = torch.randn(K, N, generator=g)
    W = torch.randn(M, N, generator=g).to(DEV).requires_grad_(True)
    xd = X.to(DEV).requires_grad_(True)
    y = pkg.ops.linear_nobias(xd, W)
    refy = X.double() @ W.detach().cpu().double().t()
    assert float((y.detach().cpu().double() - refy).abs().max()) / (float(refy.abs().max()) + 1e-12) < 1e-5
    y.backward(dY.to(DEV))
    ref = dY.double().t() @ X.double()
    err = float((W.grad.cpu().double() - ref).abs().max()) / (float(ref.abs().max()) + 1e-12)
    assert err < (2e-6 if K < 8192 else 6e-6), err      # tall-K kernel: longer fp32 chains per slice
    refx = dY.double() @ W.detach().cpu().double()
    assert float((xd.grad.cpu().double() - refx).abs().max()) / (float(refx.abs().max()) + 1e-12) < 1e-5


@pytest.mark.parametrize("K,M,N", [(100003, 256, 256), (20011, 256, 256), (9001, 132, 70), (8192, 4, 2)])
def test_gemm_tn_tall_bf16x6_is_fp32_faithful(pkg, K, M, N):
    """The tall-K product on the bf16 matrix pipe (six bf16 MFMAs per fp32 product over exact 3-way splits) against fp64, next to
    the fp32-MFMA kernel: the error must be of the same size; column sums of A as a by-product either way."""
    L = pkg._lib.lib()
    g = torch.Generator().manual_seed(K + M)
    A = (torch.randn(K, M, generator=g) * torch.exp(torch.randn(K, 1, generator=g))).to(DEV)
    A[::5] = 0
    B = torch.relu(torch.randn(K, N, generator=g)).to(DEV)
    ref = A.double().t() @ B.double()
    refc = A.double().sum(0)
    ws = pkg.ops.workspace(L.sgs_gemm_tn_workspace_bytes(K, M, N), A.device)
    err, errc = {}, {}
    try:
        for v in (0, 1):
            L.sgs_gemm_tn_set_tall_variant(v)
            C = torch.full((M, N), float("nan"), device=DEV)
            cs = torch.full((M,), float("nan"), device=DEV)
            if L.sgs_gemm_tn_can_colsum(K, M, N):
                pkg._lib.check(L.sgs_gemm_tn_colsum(A.data_ptr(), B.data_ptr(), K, M, N, C.data_ptr(), cs.data_ptr(), ws.data_ptr(),
                                                    ws.numel(), pkg.ops._stream()), "sgs_gemm_tn_colsum")
                errc[v] = float((cs.double() - refc).abs().max()) / float(refc.abs().max())
            else:
                pkg._lib.check(L.sgs_gemm_tn(A.data_ptr(), B.data_ptr(), K, M, N, C.data_ptr(), ws.data_ptr(), ws.numel(),
                                             pkg.ops._stream()), "sgs_gemm_tn")
            torch.cuda.synchronize()
            err[v] = float((C.double() - ref).abs().max()) / float(ref.abs().max())
    finally:
        L.sgs_gemm_tn_set_tall_variant(-1)
    print("max err / max|ref|:", err, "colsum:", errc)
    assert err[1] <= 2 * err[0] + 1e-7 and err[1] < 6e-6
    for v in errc:
        assert errc[v] < 2e-6


def test_gemm_tn_tall_with_column_sums(pkg):
    """sgs_gemm_tn_colsum: dW = A^T B on the tall-K kernel with colsum(A) as a by-product (d b1 of the scorer backward)."""
    L = pkg._lib.lib()
    K, M, N = 30011, 256, 256
    assert L.sgs_gemm_tn_can_colsum(K, M, N) == 1 and L.sgs_gemm_tn_can_colsum(1013, M, N) == 0
    g = torch.Generator().manual_seed(5)
    A = torch.randn(K, M, generator=g).to(DEV)
    B = torch.randn(K, N, generator=g).to(DEV)
    C = torch.empty(M, N, device=DEV)
    cs = torch.empty(M, device=DEV)
    ws = pkg.ops.workspace(L.sgs_gemm_tn_workspace_bytes(K, M, N), A.device)
    pkg._lib.check(L.sgs_gemm_tn_colsum(A.data_ptr(), B.data_ptr(), K, M, N, C.data_ptr(), cs.data_ptr(), ws.data_ptr(), ws.numel(),
                                        pkg.ops._stream()), "sgs_gemm_tn_colsum")
    ref = A.double().t() @ B.double()
    assert float((C.double() - ref).abs().max()) / float(ref.abs().max()) < 6e-6
    refc = A.double().sum(0)
    assert float((cs.double() - refc).abs().max()) / float(refc.abs().max()) < 2e-6
    with pytest.raises(RuntimeError):
        pkg._lib.check(L.sgs_gemm_tn_colsum(A.data_ptr(), B.data_ptr(), 100, M, N, C.data_ptr(), cs.data_ptr(), ws.data_ptr(), ws.numel(),
                                            pkg.ops._stream()), "sgs_gemm_tn_colsum")


@pytest.mark.parametrize("N,E,q", [(50, 600, 200), (1013, 120000, 40000), (7, 30, 29), (300, 5000, 1)])
def test_graph_filter_equals_build_of_the_compacted_edges(pkg, N, E, q):
    """sgs_graph_filter (child CSR squeezed out of the parent's CSR) == sgs_graph_build on the drawn edge list, incl. self loops
    and duplicate edges."""
    ops = pkg.ops
    g = torch.Generator().manual_seed(N + q)
    ei = torch.randint(0, N, (2, E), generator=g)
    ei[:, :5] = ei[0, :5]                                        # some self loops
    ei = ei[:, torch.argsort(ei[0] * N + ei[1], stable=True)].contiguous().to(DEV)
    p = torch.rand(E, generator=g).to(DEV)
    r = ops.sample_topq(ops.SAMPLE_LEARNED, p, None, 0.0, q, ei, seed=3, stream_id=1)
    child = ops.get_subgraph(ei, N, r)
    ref = ops.Graph(r.edge_index.clone(), N)
    torch.cuda.synchronize()
    for name in ("in_ptr", "out_ptr", "in_src", "in_eid", "out_dst", "out_eid", "loop_eid"):
        a, b = getattr(child, name), getattr(ref, name)
        n = q if name in ("in_src", "in_eid", "out_dst", "out_eid") else a.numel()
        assert torch.equal(a[:n], b[:n]), name
    assert ops.get_graph(r.edge_index, N) is child               # cached on the drawn edge list


@pytest.mark.parametrize("N,E", [(1, 0), (5, 3), (50, 700), (1013, 100000), (3, 30000), (40000, 90000)])
def test_graph_build_src_sorted_equals_graph_build(pkg, N, E):
    """sgs_graph_build_src_sorted (out-CSR = the list itself, in-CSR = one packed radix sort) == sgs_graph_build on source-sorted edge
    lists with self loops, duplicate edges and empty rows; the `unsorted` word reports a list that breaks the precondition."""
    ops, L = pkg.ops, pkg._lib.lib()
    ei, g = rand_graph(N, E, N + E)
    if E:
        ei = ei[:, torch.argsort(ei[0], stable=True)]
    ei = ei.contiguous().to(DEV)
    ref = ops.Graph(ei, N)
    ne = max(E, 1)
    arr = {k: torch.full((n_,), -7, dtype=torch.int32, device=DEV) for k, n_ in (("in_ptr", N + 1), ("out_ptr", N + 1), ("in_src", ne), ("in_eid", ne),
                                                                                 ("out_dst", ne), ("out_eid", ne), ("loop_eid", N))}
    flag = torch.full((1,), 5, dtype=torch.int32, device=DEV)
    ws = ops.workspace(L.sgs_graph_build_src_sorted_workspace_bytes(E, N), ei.device)

    def build(e):
        pkg._lib.check(L.sgs_graph_build_src_sorted(e.data_ptr(), E, N, arr["in_ptr"].data_ptr(), arr["in_src"].data_ptr(), arr["in_eid"].data_ptr(),
                                                    arr["out_ptr"].data_ptr(), arr["out_dst"].data_ptr(), arr["out_eid"].data_ptr(),
                                                    arr["loop_eid"].data_ptr(), flag.data_ptr(), ws.data_ptr(), ws.numel(), ops._stream()), "src_sorted")
        torch.cuda.synchronize()

    build(ei)
    assert int(flag) == 0
    for name, a in arr.items():
        n = E if name in ("in_src", "in_eid", "out_dst", "out_eid") else a.numel()
        assert torch.equal(a[:n], getattr(ref, name)[:n]), name
    if E > 100 and N > 1:
        bad = ei.flip(1).contiguous()
        if not bool((bad[0, 1:] >= bad[0, :-1]).all()):
            build(bad)
            assert int(flag) == 1


def test_get_subgraph_sorts_the_drawn_edges_above_its_threshold(pkg, monkeypatch):
    """get_subgraph's whole-graph path (one sort of the drawn edges) against its filter path, on the same draw."""
    ops = pkg.ops
    N, E, q = 1013, 120000, 40000
    g = torch.Generator().manual_seed(5)
    ei = torch.randint(0, N, (2, E), generator=g)
    ei[:, :5] = ei[0, :5]
    ei = ei[:, torch.argsort(ei[0] * N + ei[1], stable=True)].contiguous().to(DEV)
    p = torch.rand(E, generator=g).to(DEV)
    r = ops.sample_topq(ops.SAMPLE_LEARNED, p, None, 0.0, q, ei, seed=3, stream_id=1)
    filt = ops.get_subgraph(ei, N, r)
    monkeypatch.setattr(ops, "_SORT_SUBGRAPH_EDGES", 1000)
    r2 = ops.sample_topq(ops.SAMPLE_LEARNED, p, None, 0.0, q, ei, seed=3, stream_id=1)
    assert torch.equal(r.edge_index, r2.edge_index)
    srt = ops.get_subgraph(ei, N, r2)
    torch.cuda.synchronize()
    assert srt is not filt
    for name in ("in_ptr", "out_ptr", "in_src", "in_eid", "out_dst", "out_eid", "loop_eid"):
        a, b = getattr(srt, name), getattr(filt, name)
        n = q if name in ("in_src", "in_eid", "out_dst", "out_eid") else a.numel()
        assert torch.equal(a[:n], b[:n]), name
    # an UNSORTED parent keeps the filter (the sort path's precondition does not hold)
    perm = torch.randperm(E, generator=g).to(DEV)
    eu = ei[:, perm].contiguous()
    r3 = ops.sample_topq(ops.SAMPLE_LEARNED, p, None, 0.0, q, eu, seed=3, stream_id=1)
    child = ops.get_subgraph(eu, N, r3)
    ref = ops.Graph(r3.edge_index.clone(), N)
    torch.cuda.synchronize()
    for name in ("in_ptr", "out_ptr", "in_src", "in_eid", "out_dst", "out_eid", "loop_eid"):
        a, b = getattr(child, name), getattr(ref, name)
        n = q if name in ("in_src", "in_eid", "out_dst", "out_eid") else a.numel()
        assert torch.equal(a[:n], b[:n]), name


def test_sparse_feature_products_match_dense_and_replay_through_the_slots():
    """ops.feature_csr: bag-of-words node features (CitationFull-Cora: 0.7 % dense) run the first layers' x W^T and d W = d Y^T x as SpMMs
    over nnz(x).  (1) both products against the dense ones; (2) a layer's gradients with the sparse path on and off; (3) graph mode: the
    CSR of every partition is staged into the slots' static buffers -- two partitions with different nnz replay equal to eager training."""
    import argparse
    import sgs_gnn_amd as S
    ops = S.ops
    g = torch.Generator(device=DEV).manual_seed(4)
    N, F_, H = 2100, 2048, 64

    def feats(n, seed):
        gg = torch.Generator(device=DEV).manual_seed(seed)
        x = (torch.rand(n, F_, device=DEV, generator=gg) < 0.01).float() * torch.rand(n, F_, device=DEV, generator=gg)
        return x / x.sum(1, keepdim=True).clamp_min(1e-12)
    x = feats(N, 1)
    fc = ops.feature_csr(x, build=True)
    assert fc is not None and fc.nnz == int((x != 0).sum()) and fc.N == N and fc.F == F_
    W = torch.randn(H, F_, device=DEV, generator=g) * 0.05
    dY = torch.randn(N, H, device=DEV, generator=g)
    torch.testing.assert_close(ops._x_wt(x, W), x @ W.t(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(ops._dyt_x(dY, x, W.shape), dY.t() @ x, rtol=1e-4, atol=1e-6)
    dense = torch.randn(N, F_, device=DEV, generator=g)
    assert ops.feature_csr(dense, build=True) is None                               # N(0,1) features (Reddit): the library GEMM
    # (3) graph mode against eager on two partitions of different sizes / nnz
    crit = torch.nn.CrossEntropyLoss()
    bs = []
    for i, (n, E) in enumerate([(2100, 9000), (2080, 7000)]):
        b = S.synthetic_graph(n, E, 8, 5, seed=60 + i, device=DEV)
        b.x = feats(n, 70 + i)
        bs.append(b)

    def run(hipgraph):
        torch.manual_seed(0)
        S.fix_seeds(0)
        m = S.GNNModel(F_, 32, 5, dropout_prob=0.0, edge_mlp_type="GCN").to(DEV)
        og = S.FusedAdam([p for nme, p in m.named_parameters() if "gcn" in nme], lr=1e-2)
        oe = S.FusedAdam([p for nme, p in m.named_parameters() if "edge_prob_mlp" in nme], lr=1e-2)
        a = argparse.Namespace(device=DEV, mode="learned", pipeline="hybrid", edge_mlp_type="GCN", conditional=True, sparse_edge_mlp=True, t_init=0.7,
                               t_min=0.5, degree_bias_coef=0.3, reg1=True, reg2=True, regularizer1_coef=1.0, consist_reg_coef=0.5,
                               hybrid_checkpoint=False, sgs_hipgraph=hipgraph)
        rets = [S.train(a, ep, 3, m, og, oe, None, crit, bs, q=20_000) for ep in range(3)]       # E <= q: unsampled steps, deterministic
        return rets, m
    r_e, m_e = run(False)
    r_g, m_g = run(True)
    assert m_g._sgs_stepgraphs.fcsr_cap >= max(ops.feature_csr(b.x, build=True).nnz for b in bs)
    assert r_e == r_g
    for (k, a_), (_, b_) in zip(m_e.state_dict().items(), m_g.state_dict().items()):
        assert torch.equal(a_, b_), k
