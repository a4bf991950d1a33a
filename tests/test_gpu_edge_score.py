"""GPU parity: fused MFMA edge scorer (K1b) forward / backward vs the oracle's `_edge_score`."""
import pytest
import torch

from oracle import sgs_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    import sgs_gnn_amd
    return sgs_gnn_amd.ops


def _case(N, H, E, seed):
    g = torch.Generator().manual_seed(seed)
    codes = torch.relu(torch.randn(N, H, generator=g))          # encoder output is post-ReLU
    ei = torch.randint(0, N, (2, E), generator=g)
    if E > 5:
        ei[:, 2] = ei[0, 2]                                       # a self loop among the scored edges
    b = 1.0 / (2 * H) ** 0.5
    W1 = (torch.rand(H, 2 * H, generator=g) * 2 - 1) * b
    b1 = (torch.rand(H, generator=g) * 2 - 1) * b
    W2 = (torch.rand(1, H, generator=g) * 2 - 1) / H ** 0.5
    b2 = (torch.rand(1, generator=g) * 2 - 1) / H ** 0.5
    return codes, ei, W1, b1, W2, b2, g


def _rel(a, b):
    return float((a.double().cpu() - b.double()).abs().max()) / (float(b.double().abs().max()) + 1e-30)


@pytest.mark.parametrize("N,H,E", [(50, 16, 300), (50, 32, 129), (300, 64, 5000), (200, 128, 1000), (1013, 256, 20000), (7, 8, 1),
                                   (40, 48, 500)])
@pytest.mark.parametrize("p", [0.0, 0.3])
def test_forward_and_dense_backward(ops, N, H, E, p):
    codes, ei, W1, b1, W2, b2, g = _case(N, H, E, N + H + E)
    seed, site = 4242, 2
    keep = ops.dropout_keep(seed, site, E, H, p, DEV).cpu() if p > 0 else None
    gp = torch.randn(E, generator=g)

    leaves = [t.clone().double().requires_grad_(True) for t in (codes, W1, b1, W2, b2)]
    co, W1o, b1o, W2o, b2o = leaves
    po = O.edge_score(co[ei[0]], co[ei[1]], W1o, b1o, W2o, b2o, p, keep).squeeze(1)
    po.backward(gp.double())

    dl = [t.clone().to(DEV).requires_grad_(True) for t in (codes, W1, b1, W2, b2)]
    cd, W1d, b1d, W2d, b2d = dl
    pd = ops.edge_score(cd, W1d, b1d, W2d, b2d, ei.to(DEV), active=None, p=p, seed=seed, site=site)
    assert float((pd.detach().cpu().double() - po.detach()).abs().max()) < 2e-6          # probabilities
    pd.backward(gp.to(DEV))
    for name, a, b in zip(["dcodes", "dW1", "db1", "dW2", "db2"], dl, leaves):
        if name == "db2":     # a single signed sum of E terms: bound the error by the terms' magnitude, not the (cancelled) sum
            assert abs(float(a.grad) - float(b.grad)) < 2e-6 * float(gp.abs().sum()) / 4, name
        else:
            assert _rel(a.grad, b.grad) < 2e-5, name


def test_active_subset_backward_equals_dense_with_masked_gradient(ops):
    """Hybrid pipeline: only the q sampled edges carry gradient (training_hybrid.py:86)."""
    N, H, E, q = 400, 64, 6000, 1200
    codes, ei, W1, b1, W2, b2, g = _case(N, H, E, 9)
    eid = torch.sort(torch.randperm(E, generator=g)[:q]).values
    gq = torch.randn(q, generator=g)
    gp = torch.zeros(E)
    gp[eid] = gq
    res = []
    for use_active in (False, True):
        dl = [t.clone().to(DEV).requires_grad_(True) for t in (codes, W1, b1, W2, b2)]
        act = ops.ActiveSet()
        pd = ops.edge_score(dl[0], dl[1], dl[2], dl[3], dl[4], ei.to(DEV), active=act)
        if use_active:
            sei = ei[:, eid].to(DEV)
            act.set(eid.to(DEV), ops.Graph(sei, N))
        pd.backward(gp.to(DEV))
        res.append([t.grad.cpu() for t in dl])
    for a, b in zip(*res):
        assert _rel(a, b) < 1e-5


@pytest.mark.parametrize("p", [0.0, 0.3])
def test_production_size_forward_and_active_backward_vs_fp64_oracle(ops, p):
    """ops.edge_score forward + backward at the sizes where the library auto-selects its production kernels (E >= 65 536 and
    H % 128 == 0: bf16x6 forward; >= 65 536 active rows: bf16x6 backward core, sgs_edge_score_bwd_dfeat row GEMM, tall-K bf16x6
    weight-gradient GEMM with the bias gradient as by-product, paired endpoint reduction) against the fp64 oracle."""
    import sgs_gnn_amd as S
    L = S._lib.lib()
    L.sgs_edge_score_set_variant(-1)
    L.sgs_edge_score_set_bwd_variant(-1)
    L.sgs_gemm_tn_set_tall_variant(-1)
    N, H, E, q = 1013, 256, 150_001, 70_003
    codes, ei, W1, b1, W2, b2, g = _case(N, H, E, 2024)
    seed, site = 777, 2
    eid = torch.sort(torch.randperm(E, generator=g)[:q]).values
    gq = torch.randn(q, generator=g)
    keep = ops.dropout_keep(seed, site, E, H, p, DEV).cpu() if p > 0 else None

    # fp64 oracle: forward over all E edges (chunked: [E,2H] in fp64 is 0.6 GB), backward through the q active rows only
    leaves = [t.clone().double().requires_grad_(True) for t in (codes, W1, b1, W2, b2)]
    co, W1o, b1o, W2o, b2o = leaves
    with torch.no_grad():
        po_all = torch.cat([O.edge_score(co[ei[0, a:a + 32768]], co[ei[1, a:a + 32768]], W1o, b1o, W2o, b2o, p,
                                         None if keep is None else keep[a:a + 32768]).squeeze(1) for a in range(0, E, 32768)])
    sub = ei[:, eid]
    po = O.edge_score(co[sub[0]], co[sub[1]], W1o, b1o, W2o, b2o, p, None if keep is None else keep[eid]).squeeze(1)
    po.backward(gq.double())

    dl = [t.clone().to(DEV).requires_grad_(True) for t in (codes, W1, b1, W2, b2)]
    act = ops.ActiveSet()
    pd = ops.edge_score(dl[0], dl[1], dl[2], dl[3], dl[4], ei.to(DEV), active=act, p=p, seed=seed, site=site)
    assert float((pd.detach().cpu().double() - po_all).abs().max()) < 2e-6
    act.set(eid.to(DEV), ops.Graph(sub.to(DEV), N))
    gp = torch.zeros(E)
    gp[eid] = gq
    pd.backward(gp.to(DEV))
    for name, a, b in zip(["dcodes", "dW1", "db1", "dW2", "db2"], dl, leaves):
        if name == "db2":
            assert abs(float(a.grad) - float(b.grad)) < 2e-6 * float(gq.abs().sum()) / 4, name
        else:
            assert _rel(a.grad, b.grad) < 2e-5, (name, _rel(a.grad, b.grad))


@pytest.mark.parametrize("N,H,p", [(777, 256, 0.3), (777, 128, 0.0), (777, 256, 0.0), (70_001, 128, 0.3)])
def test_mask_backward_equals_dense_backward(ops, N, H, p):
    """Three forms of the backward on the same inputs: (a) the forward keeps the ReLU x dropout mask and the backward recomputes nothing
    (dz from p, d fc2.weight from the consumers' parts), (b) the core recomputes the hidden layer and writes dv as one bit per entry,
    (c) the dense fp32-dv path.  The same five gradients to fp32 rounding: (a) and (b) differ from (c) only in where the row and column
    factors dz, w2 / (1 - p) are multiplied in, (a) from (b) in how d fc2.weight is summed."""
    E, q = 140_000, 66_000                     # (N > 65 536: the row-block reductions on a whole graph)
    codes, ei, W1, b1, W2, b2, g = _case(N, H, E, 99)
    eid = torch.sort(torch.randperm(E, generator=g)[:q]).values
    gp = torch.zeros(E)
    gp[eid] = torch.randn(q, generator=g)
    sub = ei[:, eid]
    grads, probs = {}, {}
    for form, (fwd_mask, mask) in {"kept": (True, True), "bits": (False, True), "dense": (False, False)}.items():
        ops._fwd_mask, ops._mask_backward = fwd_mask, mask
        try:
            dl = [t.clone().to(DEV).requires_grad_(True) for t in (codes, W1, b1, W2, b2)]
            act = ops.ActiveSet()
            pd = ops.edge_score(dl[0], dl[1], dl[2], dl[3], dl[4], ei.to(DEV), active=act, p=p, seed=5, site=2)
            act.set(eid.to(DEV), ops.Graph(sub.to(DEV), N))
            pd.backward(gp.to(DEV))
            grads[form] = [t.grad.detach().cpu() for t in dl]
            probs[form] = pd.detach().cpu()
        finally:
            ops._fwd_mask, ops._mask_backward = True, True
    assert torch.equal(probs["kept"], probs["dense"])                      # the mask-keeping forward scores bit for bit as the plain one
    for form in ("kept", "bits"):
        for name, a, b in zip(["dcodes", "dW1", "db1", "dW2", "db2"], grads[form], grads["dense"]):
            assert bool(torch.isfinite(a).all()), (form, name)
            assert _rel(a, b) < 3e-6, (form, name, _rel(a, b))


@pytest.mark.parametrize("H,p", [(256, 0.3), (128, 0.0)])
def test_forward_mask_prep_and_dw2_parts_entry_points(ops, H, p):
    """The entry points of the no-recompute backward, one by one, through the C ABI:
    sgs_edge_score_fwd_mask -- scores bit-identical to sgs_edge_score_fwd, mask bits == [dropout(relu(v)) > 0] from the fp64 oracle's
    pre-activations (entries with |v| < 1e-5 excepted: the sign of a rounding-level v may differ);
    sgs_edge_score_bwd_prep -- dz = gp p (1 - p), the active rows' mask, feat = x_s * x_d, exactly;
    sgs_edge_score_dw2_from_parts -- against sum_e dz_e hidden[e, :] in fp64 from parts built in fp64."""
    import sgs_gnn_amd as S
    L = S._lib.lib()
    N, E, q = 500, 70_000, 66_000
    codes, ei, W1, b1, W2, b2, g = _case(N, H, E, 31)
    seed, site = 11, 2
    d = lambda t: t.to(DEV).contiguous()
    codes_d, ei_d, W1_d, b1_d, w2_d, b2_d = d(codes), d(ei), d(W1), d(b1), d(W2.reshape(-1)), d(b2)
    U_d = (codes_d @ W1_d[:, H:].t()).contiguous()
    ws = ops.workspace(L.sgs_edge_score_workspace_bytes(N, H, E), codes_d.device)
    p_plain = torch.empty(E, device=DEV)
    p_mask = torch.empty(E, device=DEV)
    bits = torch.zeros(E, H // 32, dtype=torch.int32, device=DEV)
    st = ops._stream()
    S._lib.check(L.sgs_edge_score_fwd(codes_d.data_ptr(), U_d.data_ptr(), N, H, ei_d.data_ptr(), E, 0, W1_d.data_ptr(), b1_d.data_ptr(), w2_d.data_ptr(),
                                      b2_d.data_ptr(), p, seed, site, p_plain.data_ptr(), ws.data_ptr(), ws.numel(), st), "fwd")
    S._lib.check(L.sgs_edge_score_fwd_mask(codes_d.data_ptr(), U_d.data_ptr(), N, H, ei_d.data_ptr(), E, 0, None, 0, None, W1_d.data_ptr(), b1_d.data_ptr(),
                                           w2_d.data_ptr(), b2_d.data_ptr(), p, seed, site, p_mask.data_ptr(), bits.data_ptr(), ws.data_ptr(), ws.numel(), st),
                 "fwd_mask")
    assert torch.equal(p_plain, p_mask)
    # mask bits vs the oracle's pre-activations
    keep = ops.dropout_keep(seed, site, E, H, p, DEV).cpu() if p > 0 else torch.ones(E, H, dtype=torch.bool)
    x, y = codes.double()[ei[0]], codes.double()[ei[1]]
    v = torch.cat([x * y, x - y], 1) @ W1.double().t() + b1.double()
    want = (v > 0) & keep
    got = ((bits.cpu().view(E, H // 32, 1) >> torch.arange(32).view(1, 1, 32)) & 1).bool().view(E, H)
    clear = v.abs() > 1e-5
    assert bool((got == want)[clear].all()) and float(clear.double().mean()) > 0.999
    # prep
    eid = torch.sort(torch.randperm(E, generator=g)[:q]).values
    gq = torch.randn(q, generator=g)
    dz = torch.empty(q, device=DEV)
    bact = torch.empty(q, H // 32, dtype=torch.int32, device=DEV)
    feat = torch.empty(q, H, device=DEV)
    eid_d, gq_d = d(eid), d(gq)                  # (held in variables: a temporary's storage may be handed to the next allocation)
    S._lib.check(L.sgs_edge_score_bwd_prep(codes_d.data_ptr(), N, H, ei_d.data_ptr(), E, eid_d.data_ptr(), q, gq_d.data_ptr(), p_mask.data_ptr(),
                                           bits.data_ptr(), dz.data_ptr(), bact.data_ptr(), feat.data_ptr(), st), "prep")
    torch.cuda.synchronize()
    pc = p_mask.cpu()[eid]
    assert torch.equal(dz.cpu(), gq * pc * (1.0 - pc))
    assert torch.equal(bact.cpu(), bits.cpu()[eid])
    assert torch.equal(feat.cpu(), codes[ei[0, eid]] * codes[ei[1, eid]])
    # d fc2.weight from parts (built here in fp64 from the kernel's own mask)
    scale = 1.0 / (1.0 - p)
    m = got[eid].double()
    dzd = dz.cpu().double()
    T = (m * dzd[:, None]).t() @ (codes.double()[ei[0, eid]] * codes.double()[ei[1, eid]])
    c = (m * dzd[:, None]).sum(0)
    R = torch.zeros(N, H, dtype=torch.float64)
    R.index_add_(0, ei[0, eid], m * dzd[:, None])
    R.index_add_(0, ei[1, eid], -(m * dzd[:, None]))
    dw2 = torch.empty(H, device=DEV)
    T_d, R_d, c_d = d(T.float()), d(R.float()), d(c.float())
    S._lib.check(L.sgs_edge_score_dw2_from_parts(W1_d.data_ptr(), T_d.data_ptr(), U_d.data_ptr(), R_d.data_ptr(), b1_d.data_ptr(), c_d.data_ptr(), N, H,
                                                 p, dw2.data_ptr(), st), "dw2")
    torch.cuda.synchronize()
    hidden = torch.where(got[eid], v[eid] * scale, torch.zeros((), dtype=torch.float64))
    ref = (dzd[:, None] * hidden).sum(0)
    assert _rel(dw2, ref) < 2e-5, _rel(dw2, ref)


@pytest.mark.parametrize("H,p", [(256, 0.3), (128, 0.0)])
def test_paired_forward_equals_plain_forward_bitwise(ops, H, p):
    """sgs_edge_score_fwd_paired (only the canonical edge of every (s -> d), (d -> s) pair runs the H x H contraction; both scores
    are finished from one set of accumulators) against sgs_edge_score_fwd on the same inputs: bit-identical p for every edge, on
    an undirected graph stored both ways with extras that must stay unmated or pair off one to one -- self loops, one-directional
    edges, duplicate edges -- and through autograd (the backward does not depend on which forward ran)."""
    import sgs_gnn_amd as S
    N = 700
    b = S.synthetic_graph(N, 90_000, 8, 3, seed=4, device=DEV)                    # symmetric, coalesced, row-sorted
    g = torch.Generator().manual_seed(8)
    extra = torch.randint(0, N, (2, 3000), generator=g)                             # one-directional edges (some duplicate existing ones)
    loops = torch.arange(0, 50).repeat(2, 1)
    dup = b.edge_index[:, :500].cpu()                                               # exact duplicates of mated edges
    ei = torch.cat([b.edge_index.cpu(), extra, loops, dup], dim=1)
    ei = ei[:, torch.argsort(ei[0] * N + ei[1], stable=True)].contiguous().to(DEV)
    E = ei.shape[1]
    codes, _, W1, b1, W2, b2, _ = _case(N, H, 10, 5)
    canon, mate = ops.get_pairs(ei, N, build=True)
    m = mate[:E].long().cpu()
    ar = torch.arange(E)
    paired = m >= 0
    assert torch.equal(m[m[paired]], ar[paired])                                    # an involution on the mated edges
    assert torch.equal(ei.cpu()[:, m[paired]], ei.cpu()[:, paired].flip(0))         # ... onto the reverse edge
    assert int(paired.sum()) >= b.edge_index.shape[1] and not bool(paired[(ei[0] == ei[1]).cpu()].any())
    assert canon.numel() == E - int(paired.sum()) // 2
    seed, site = 31, 2
    outs = []
    for pr in (None, (canon, mate)):
        dl = [t.clone().to(DEV).requires_grad_(True) for t in (codes, W1, b1, W2, b2)]
        pd = ops.edge_score(dl[0], dl[1], dl[2], dl[3], dl[4], ei, p=p, seed=seed, site=site, pairs=pr)
        pd.sum().backward()
        outs.append((pd.detach().clone(), [t.grad.clone() for t in dl]))
    assert torch.equal(outs[0][0], outs[1][0])
    for ga, gb in zip(outs[0][1], outs[1][1]):
        assert torch.equal(ga, gb)
    # and against the oracle, as every other forward variant
    keep = ops.dropout_keep(seed, site, E, H, p, DEV).cpu() if p > 0 else None
    eic = ei.cpu()
    po = O.edge_score(codes[eic[0]].double(), codes[eic[1]].double(), W1.double(), b1.double(), W2.double(), b2.double(), p, keep).squeeze(1)
    assert float((outs[1][0].cpu().double() - po).abs().max()) < 2e-6


def test_scorer_tail_and_probability_range(ops):
    N, H, E = 1013, 256, 100001          # E not a multiple of the 128-edge tile
    codes, ei, W1, b1, W2, b2, _ = _case(N, H, E, 3)
    pd = ops.edge_score(codes.to(DEV), W1.to(DEV), b1.to(DEV), W2.to(DEV), b2.to(DEV), ei.to(DEV))
    po = O.edge_score(codes[ei[0]], codes[ei[1]], W1, b1, W2, b2).squeeze(1)
    assert pd.shape == (E,) and float(pd.min()) > 0 and float(pd.max()) < 1
    assert float((pd.cpu() - po).abs().max()) < 2e-6


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("N,H,E,p", [(300, 64, 5001, 0.0), (1013, 256, 30011, 0.3), (200, 128, 77, 0.3), (500, 128, 9000, 0.0)])
def test_all_forward_variants_match_oracle(ops, variant, N, H, E, p):
    """LDS-tiled (0), register-streaming 32-edge (1) / 64-edge (3) wave tiles, weight-stationary persistent (2) and
    bf16x6 split (4; falls back to 3 unless H % 128 == 0) forward kernels."""
    import sgs_gnn_amd as S
    L = S._lib.lib()
    codes, ei, W1, b1, W2, b2, g = _case(N, H, E, 31 + H)
    seed, site = 99, 2
    keep = ops.dropout_keep(seed, site, E, H, p, DEV).cpu() if p > 0 else None
    po = O.edge_score(codes[ei[0]].double(), codes[ei[1]].double(), W1.double(), b1.double(), W2.double(), b2.double(), p, keep).squeeze(1)
    try:
        L.sgs_edge_score_set_variant(variant)
        pd = ops.edge_score(codes.to(DEV), W1.to(DEV), b1.to(DEV), W2.to(DEV), b2.to(DEV), ei.to(DEV), p=p, seed=seed, site=site)
        torch.cuda.synchronize()
    finally:
        L.sgs_edge_score_set_variant(-1)
    assert float((pd.cpu().double() - po).abs().max()) < 2e-6


@pytest.mark.parametrize("H", [128, 256])
def test_bf16x6_split_is_fp32_faithful(ops, H):
    """Variant 4 evaluates the fp32 contraction as six bf16 MFMAs over exact 3-way operand splits.  Its error against an fp64
    evaluation must be of the same size as that of the fp32 MFMA kernel (variant 3) -- i.e. fp32 rounding, not bf16 rounding --
    on operands with a wide dynamic range (codes up to ~40, logits of a few units)."""
    import sgs_gnn_amd as S
    L = S._lib.lib()
    N, E = 700, 50000
    codes, ei, W1, b1, W2, b2, g = _case(N, H, E, 77 + H)
    codes = codes * torch.exp(torch.randn(N, 1, generator=g))         # per-node scale: products span several binades
    W2 = W2 * 4
    po = O.edge_score(codes[ei[0]].double(), codes[ei[1]].double(), W1.double(), b1.double(), W2.double(), b2.double()).squeeze(1)
    assert float(po.std()) > 0.05                                      # the case is not saturated
    err = {}
    try:
        for v in (3, 4):
            L.sgs_edge_score_set_variant(v)
            pd = ops.edge_score(codes.to(DEV), W1.to(DEV), b1.to(DEV), W2.to(DEV), b2.to(DEV), ei.to(DEV))
            torch.cuda.synchronize()
            err[v] = float((pd.cpu().double() - po).abs().max())
    finally:
        L.sgs_edge_score_set_variant(-1)
    print("max |p - p_fp64|:", err)
    assert err[4] <= 2 * err[3] + 6e-8


@pytest.mark.parametrize("other", [3, 4])
@pytest.mark.parametrize("H,p", [(256, 0.3), (128, 0.0)])
def test_backward_core_stream64_matches_lds_tiled(ops, H, p, other):
    """The backward core on the 64-edge streaming loop (3) and on the bf16x6 loop (4), forced here, against the LDS-tiled
    core: dv, feat, dz and the column sums of the per-tile dz * hidden partials."""
    import sgs_gnn_amd as S
    L = S._lib.lib()
    N, E, n = 1013, 60000, 40003
    g = torch.Generator().manual_seed(H)
    codes = torch.relu(torch.randn(N, H, generator=g)).to(DEV)
    ei = torch.randint(0, N, (2, E), generator=g).to(DEV)
    eid = torch.sort(torch.randperm(E, generator=g)[:n]).values.to(DEV)
    W1 = (torch.randn(H, 2 * H, generator=g) / (2 * H) ** 0.5).to(DEV)
    b1, w2, b2 = (torch.randn(H, generator=g) * 0.1).to(DEV), (torch.randn(H, generator=g) / H ** 0.5).to(DEV), torch.zeros(1, device=DEV)
    U = (codes @ W1[:, H:].t()).contiguous()
    gp = torch.randn(n, generator=g).to(DEV)
    tile = L.sgs_edge_score_bwd_tile()
    outs = []
    try:
        for variant in (0, other):
            L.sgs_edge_score_set_bwd_variant(variant)
            dv, feat = torch.full((n, H), float("nan"), device=DEV), torch.full((n, H), float("nan"), device=DEV)
            hdz = torch.full(((n + tile - 1) // tile, H), float("nan"), device=DEV)
            dz = torch.full((n,), float("nan"), device=DEV)
            ws = ops.workspace(L.sgs_edge_score_workspace_bytes(N, H, 0), codes.device)
            S._lib.check(L.sgs_edge_score_bwd_core(codes.data_ptr(), U.data_ptr(), N, H, ei.data_ptr(), E, 0, eid.data_ptr(), n, gp.data_ptr(),
                                                   W1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), p, 5, 2, dv.data_ptr(), hdz.data_ptr(),
                                                   dz.data_ptr(), feat.data_ptr(), ws.data_ptr(), ws.numel(), ops._stream()), "bwd_core")
            torch.cuda.synchronize()
            outs.append((dv.cpu(), feat.cpu(), dz.cpu(), hdz.sum(0).cpu()))
    finally:
        L.sgs_edge_score_set_bwd_variant(-1)
    (dv0, f0, dz0, h0), (dv1, f1, dz1, h1) = outs
    assert torch.isfinite(dv1).all() and torch.isfinite(f1).all() and torch.isfinite(dz1).all() and torch.isfinite(h1).all()
    assert torch.equal(f0, f1)                                           # x_s * x_d: the same single product either way
    torch.testing.assert_close(dz1, dz0, rtol=2e-4, atol=1e-6)
    # dv = dz w2 relu'(v) keep: the two cores sum v in different orders, so a hidden unit whose pre-activation is within rounding of 0 may
    # be on in one and off in the other (one entry in 10 M with this seed's masks): a handful of such flips is not an error, more would be
    bad = (dv1 - dv0).abs() > (1e-6 + 2e-4 * dv0.abs())
    assert int(bad.sum()) <= 8, int(bad.sum())
    torch.testing.assert_close(h1, h0, rtol=2e-4, atol=2e-3)


@pytest.mark.parametrize("H,n", [(256, 100003), (128, 70000), (256, 65)])
def test_bwd_dfeat_row_gemm_is_fp32_faithful(ops, H, n):
    """dfeat = dv . W1[:, :H] on the bf16x6 loop (row-GEMM mode): error against fp64 no larger than a plain fp32 GEMM's."""
    import sgs_gnn_amd as S
    L = S._lib.lib()
    assert L.sgs_edge_score_bwd_dfeat_supported(H) == 1 and L.sgs_edge_score_bwd_dfeat_supported(64) == 0
    g = torch.Generator().manual_seed(H + n)
    dv = (torch.randn(n, H, generator=g) * torch.exp(torch.randn(n, 1, generator=g))).to(DEV)
    dv[::7] = 0                                                          # rows the gate / ReLU zeroed
    W1 = (torch.randn(H, 2 * H, generator=g) / (2 * H) ** 0.5).to(DEV)
    out = torch.full((n, H), float("nan"), device=DEV)
    ws = ops.workspace(L.sgs_edge_score_workspace_bytes(0, H, 0), dv.device)
    S._lib.check(L.sgs_edge_score_bwd_dfeat(dv.data_ptr(), n, H, W1.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel(), ops._stream()),
                 "bwd_dfeat")
    torch.cuda.synchronize()
    ref64 = dv.double() @ W1[:, :H].double()
    ref32 = dv @ W1[:, :H]
    scale = float(ref64.abs().max())
    e_new = float((out.double() - ref64).abs().max()) / scale
    e_lib = float((ref32.double() - ref64).abs().max()) / scale
    print("max err / max|ref|: bf16x6", e_new, "library fp32 GEMM", e_lib)
    assert torch.isfinite(out).all()
    assert e_new <= 2 * e_lib + 1e-7


def _sorted_case(N, H, E, q, seed, hub=6000):
    """A row-sorted edge list (PyG's layout) with a hub source, a node range without out-edges, and a drawn subset of q edges."""
    codes, ei, W1, b1, W2, b2, g = _case(N, H, E, seed)
    ei = ei.clone()
    ei[0, :hub] = 3                                          # a hub: its drawn out-edges span hundreds of 32-row tiles
    ei[0, ei[0] % 7 == 5] = 2                                # ... and many sources without a single out-edge
    ei = ei[:, torch.argsort(ei[0] * N + ei[1], stable=True)].contiguous()
    eid = torch.sort(torch.randperm(E, generator=g)[:q]).values
    return codes, ei, W1, b1, W2, b2, g, eid


@pytest.mark.parametrize("N,H,p,q", [(777, 256, 0.3, 66_000), (777, 128, 0.0, 70_001), (1013, 256, 0.3, 100_003), (70_001, 128, 0.3, 66_000)])
def test_fused_backward_equals_unfused_backward(ops, N, H, p, q):
    """The fused form of the no-recompute backward (active rows sorted by source: neither feat nor dfeat is materialised, the by-source
    half of d codes is reduced inside the dfeat contraction) against the unfused form and the dense-dv form on the same inputs."""
    E = 140_000
    codes, ei, W1, b1, W2, b2, g, eid = _sorted_case(N, H, E, q, 123)
    gp = torch.zeros(E)
    gp[eid] = torch.randn(q, generator=g)
    sub = ei[:, eid]
    grads = {}
    for form, (fused, mask) in {"fused": (True, True), "unfused": (False, True), "dense": (False, False)}.items():
        ops._fused_backward, ops._mask_backward = fused, mask
        try:
            dl = [t.clone().to(DEV).requires_grad_(True) for t in (codes, W1, b1, W2, b2)]
            act = ops.ActiveSet()
            ei_d = ei.to(DEV)
            assert ops.src_sorted(ei_d)
            pd = ops.edge_score(dl[0], dl[1], dl[2], dl[3], dl[4], ei_d, active=act, p=p, seed=5, site=2)
            act.set(eid.to(DEV), ops.Graph(sub.to(DEV), N))
            pd.backward(gp.to(DEV))
            grads[form] = [t.grad.detach().cpu() for t in dl]
        finally:
            ops._fused_backward, ops._mask_backward = True, True
    for form in ("fused", "unfused"):
        for name, a, b in zip(["dcodes", "dW1", "db1", "dW2", "db2"], grads[form], grads["dense"]):
            assert bool(torch.isfinite(a).all()), (form, name)
            # (d b2 is ONE sum of q signed terms: its relative error is the summation order's, not the kernels')
            assert _rel(a, b) < (2e-5 if name == "db2" else 3e-6), (form, name, _rel(a, b))
    # the weight gradient gathers exactly the products the materialised feat held; the shared-operand kernel (round 3) sums them in
    # another order than the per-wave-slice kernel of the unfused form
    assert _rel(grads["fused"][1][:, :H], grads["unfused"][1][:, :H]) < 2e-6
    assert _rel(grads["fused"][2], grads["unfused"][2]) < 2e-6 and _rel(grads["fused"][4], grads["unfused"][4]) < 2e-5


@pytest.mark.parametrize("H,p,n", [(256, 0.3, 66_003), (128, 0.0, 70_000), (256, 0.0, 95)])
def test_fused_dfeat_entry_point_by_source_partials_and_G(ops, H, p, n):
    """sgs_edge_score_bwd_dfeat_fused through the C ABI: G == dfeat * codes[src] bit for bit (dfeat from sgs_edge_score_bwd_dfeat_bits), and
    every node's run-end rows of `opart` add up to its by-source sum of dfeat * codes[dst]; sgs_edge_score_bwd_reduce_fused ==
    sgs_endpoint_reduce_pair_bits to fp32 rounding."""
    import sgs_gnn_amd as S
    L = S._lib.lib()
    N = 600
    g = torch.Generator().manual_seed(H + n)
    codes = torch.relu(torch.randn(N, H, generator=g)).to(DEV)
    src = torch.randint(0, N, (n,), generator=g)
    src[: min(n, 3000) // 2] = 0                                       # a long run at the start
    src = torch.sort(src).values
    dst = torch.randint(0, N, (n,), generator=g)
    sub = torch.stack([src, dst]).to(DEV)
    sd = sub.t().contiguous().to(torch.int32)
    W1 = (torch.randn(H, 2 * H, generator=g) / (2 * H) ** 0.5).to(DEV)
    w2 = (torch.randn(H, generator=g) / H ** 0.5).to(DEV)
    dz = torch.randn(n, generator=g).to(DEV)
    bits = torch.randint(-2**31, 2**31 - 1, (n, H // 32), generator=g, dtype=torch.int64).to(torch.int32).to(DEV)
    st = ops._stream()
    ws = ops.workspace(L.sgs_edge_score_workspace_bytes(0, H, 0), codes.device)
    dfeat = torch.full((n, H), float("nan"), device=DEV)
    S._lib.check(L.sgs_edge_score_bwd_dfeat_bits(bits.data_ptr(), dz.data_ptr(), n, H, W1.data_ptr(), w2.data_ptr(), p, dfeat.data_ptr(), ws.data_ptr(),
                                                 ws.numel(), st), "dfeat_bits")
    G = torch.full((n, H), float("nan"), device=DEV)
    rows = L.sgs_edge_score_bwd_fused_opart_rows(n, N)
    opart = torch.full((rows, H), float("nan"), device=DEV)
    S._lib.check(L.sgs_edge_score_bwd_dfeat_fused(bits.data_ptr(), dz.data_ptr(), sd.data_ptr(), codes.data_ptr(), n, N, H, W1.data_ptr(), w2.data_ptr(), p,
                                                  G.data_ptr(), opart.data_ptr(), ws.data_ptr(), ws.numel(), st), "dfeat_fused")
    torch.cuda.synchronize()
    assert torch.equal(G, dfeat * codes[sub[0]])
    # run-end rows: the last row of every source inside every 32-row tile, slot (r >> 5) + src
    r = torch.arange(n, device=DEV)
    is_end = torch.ones(n, dtype=torch.bool, device=DEV)
    is_end[:-1] = (sub[0, 1:] != sub[0, :-1]) | (r[:-1] % 32 == 31)
    slots = (r[is_end] >> 5) + sub[0, is_end]
    assert slots.unique().numel() == slots.numel() and int(slots.max()) < rows
    got = torch.zeros(N, H, device=DEV, dtype=torch.float64)
    got.index_add_(0, sub[0, is_end], opart[slots].double())
    want = torch.zeros(N, H, device=DEV, dtype=torch.float64)
    want.index_add_(0, sub[0], (dfeat * codes[sub[1]]).double())
    assert bool(torch.isfinite(opart[slots]).all())
    assert _rel(got.float(), want.float().cpu()) < 2e-6
    # the reductions
    graph = ops.Graph(sub, N)
    assert torch.equal(graph.out_eid[:n].long(), torch.arange(n, device=DEV))        # rows sorted by source: the out-CSR is the identity
    oc1, ou1, ur1 = (torch.full((N, H), float("nan"), device=DEV) for _ in range(3))
    oc2, ou2, ur2 = (torch.full((N, H), float("nan"), device=DEV) for _ in range(3))
    S._lib.check(L.sgs_endpoint_reduce_pair_bits(dfeat.data_ptr(), bits.data_ptr(), dz.data_ptr(), w2.data_ptr(), p, codes.data_ptr(), N, H, n,
                                                 graph.in_ptr.data_ptr(), graph.in_src.data_ptr(), graph.in_eid.data_ptr(), graph.out_ptr.data_ptr(),
                                                 graph.out_dst.data_ptr(), graph.out_eid.data_ptr(), oc1.data_ptr(), ou1.data_ptr(), ur1.data_ptr(), st),
                 "reduce_pair_bits")
    S._lib.check(L.sgs_edge_score_bwd_reduce_fused(G.data_ptr(), opart.data_ptr(), bits.data_ptr(), dz.data_ptr(), w2.data_ptr(), p, N, H, n,
                                                   graph.in_ptr.data_ptr(), graph.in_eid.data_ptr(), graph.out_ptr.data_ptr(), oc2.data_ptr(), ou2.data_ptr(),
                                                   ur2.data_ptr(), st), "reduce_fused")
    torch.cuda.synchronize()
    assert _rel(oc2, oc1.cpu()) < 3e-6 and _rel(ou2, ou1.cpu()) < 3e-6 and _rel(ur2, ur1.cpu()) < 3e-6


@pytest.mark.parametrize("N,H,E,q,p", [(500, 256, 70_001, 20_000, 0.3), (300, 64, 9_000, None, 0.3), (400, 128, 66_000, 66_000, 0.5), (200, 32, 3000, 500, 0.0)])
def test_endpoint_dropout_scorer_vs_fp64_oracle(ops, N, H, E, q, p):
    """EdgeProbMLP's scorer with per-(edge, endpoint) dropout inside the kernels (ops.edge_score_epd: sgs_edge_score_epd_fwd / _bwd_core /
    _reduce) against the fp64 oracle fed the SAME masks (sgs_dropout_keep): forward for every edge, backward over an active subset
    (q rows; None = dense over all E) -- d A, d fc1.weight (both halves), d fc1.bias, d fc2.weight, d fc2.bias."""
    from sgs_gnn_amd.model import SITE_MLP_X, SITE_MLP_Y, SITE_SCORE
    codes, ei, W1, b1, W2, b2, g = _case(N, H, E, 200 + H)
    sx, sy, ss = 11, 12, 13
    if p > 0:
        kx, ky, kh = (ops.dropout_keep(sd, site, E, H, p, DEV).cpu() for sd, site in ((sx, SITE_MLP_X), (sy, SITE_MLP_Y), (ss, SITE_SCORE)))
    else:
        kx = ky = torch.ones(E, H, dtype=torch.bool)
        kh = None
    Ao = codes.clone().double().requires_grad_(True)
    Po = [t.clone().double().requires_grad_(True) for t in (W1, b1, W2, b2)]
    xm = Ao[ei[0]] * kx / (1 - p)
    ym = Ao[ei[1]] * ky / (1 - p)
    po = O.edge_score(xm, ym, Po[0], Po[1], Po[2], Po[3], p, kh).squeeze(1)
    dl = [t.clone().to(DEV).requires_grad_(True) for t in (codes, W1, b1, W2, b2)]
    act = ops.ActiveSet() if q is not None else None
    ei_d = ei.to(DEV)
    pd = ops.edge_score_epd(dl[0], dl[1], dl[2], dl[3], dl[4], ei_d, active=act, p=p, seed=ss, site=SITE_SCORE, p_ep=p, seed_x=sx,
                            site_x=SITE_MLP_X, seed_y=sy, site_y=SITE_MLP_Y)
    assert float((pd.detach().cpu().double() - po.detach()).abs().max()) < 2e-6
    gp = torch.zeros(E)
    if q is not None:
        eid = torch.sort(torch.randperm(E, generator=g)[:q]).values
        gp[eid] = torch.randn(q, generator=g)
        act.set(eid.to(DEV), ops.Graph(ei[:, eid].to(DEV), N))
    else:
        gp = torch.randn(E, generator=g)
    # an edge with a hidden unit whose pre-activation is within fp32 rounding of 0 may have that unit on in fp32 and off in fp64 (one such
    # unit -- |v| = 6e-8 -- among 576 000 with one of the seeds): ReLU' is not continuous there, so those edges carry no gradient here
    with torch.no_grad():
        vpre = torch.cat([xm * ym, xm - ym], 1) @ Po[0].t() + Po[1]
        gp[(vpre.abs() < 1e-5).any(1)] = 0.0
    po.backward(gp.double())
    pd.backward(gp.to(DEV))
    for name, a, b in zip(["dA", "dW1", "db1", "dW2", "db2"], [t.grad for t in dl], [Ao.grad, Po[0].grad, Po[1].grad, Po[2].grad, Po[3].grad]):
        assert bool(torch.isfinite(a).all()), name
        assert _rel(a, b.reshape(a.shape)) < 2e-5, (name, _rel(a, b.reshape(a.shape)))


def test_paired_forward_is_run_to_run_deterministic_at_arxiv_size(ops):
    """Config 4's partition size (n = 33 869: the codes table is 34 MB, not L2-resident; E = 463 k): the paired forward -- plain and
    mask-keeping -- must give the same bits on every launch.  A round-3 build with SLP-packed fp32 epilogue arithmetic (v_pk_add_f32 /
    v_pk_mul_f32) did not (sgs-gnn_amd/build.py: -fno-slp-vectorize); this is the regression test for that."""
    import sgs_gnn_amd as S
    n, H = 33_869, 256
    b = S.synthetic_graph(n, 463_000, 8, 5, seed=300, train_frac=0.2, power=0.6, device=DEV)
    g = torch.Generator(device=DEV).manual_seed(1)
    codes = torch.relu(torch.randn(n, H, device=DEV, generator=g))
    W1 = torch.randn(H, 2 * H, device=DEV, generator=g) / (2 * H) ** 0.5
    b1 = torch.randn(H, device=DEV, generator=g) * 0.05
    W2 = torch.randn(1, H, device=DEV, generator=g) / H ** 0.5
    b2 = torch.zeros(1, device=DEV)
    pairs = ops.get_pairs(b.edge_index, n, build=True)
    for grad in (False, True):
        cd = codes.clone().requires_grad_(grad)                      # grad: the mask-keeping forward of a training step
        with torch.set_grad_enabled(grad):
            p0 = ops.edge_score(cd, W1, b1, W2, b2, b.edge_index, pairs=pairs, p=0.3, seed=3, site=2).detach().clone()
            for it in range(25):
                p1 = ops.edge_score(cd, W1, b1, W2, b2, b.edge_index, pairs=pairs, p=0.3, seed=3, site=2).detach()
                assert torch.equal(p1, p0), (grad, it, int((p1 != p0).sum()))


@pytest.mark.parametrize("K,H,N", [(100_000, 256, 1013), (70_001, 128, 500), (8_200, 256, 64), (131_072, 256, 33_869)])
def test_gather_weight_gradient_shared_operand_kernel(K, H, N):
    """sgs_gemm_tn_mask_gather, shared-operand kernel (the four waves of a K-group split a step's rows once and exchange fragments through
    LDS) against fp64 and against the per-wave-slice kernel of round 2: C (strided), C_raw, column sums (raw and scaled), dz sum.
    Ragged K (not a multiple of the step), both hidden widths, a table larger than L2."""
    import sgs_gnn_amd as S
    ops, L = S.ops, S._lib.lib()
    g = torch.Generator(device=DEV).manual_seed(K + H)
    codes = torch.relu(torch.randn(N, H, device=DEV, generator=g))
    sd = torch.randint(0, N, (K, 2), device=DEV, generator=g, dtype=torch.int32)
    bits = torch.randint(-2**31, 2**31 - 1, (K, H // 32), device=DEV, generator=g, dtype=torch.int64).to(torch.int32)
    dz = torch.randn(K, device=DEV, generator=g)
    w2 = torch.randn(H, device=DEV, generator=g)
    scale = 1.0 / 0.7
    assert L.sgs_gemm_tn_mask_supported(K, H, H)
    ws = ops.workspace(L.sgs_gemm_tn_workspace_bytes(K, H, H), codes.device)

    def run(shared, slabs=0):
        L.sgs_gemm_tn_set_gather_variant(shared, slabs)
        C = torch.full((H, 2 * H), 7.0, device=DEV)
        cs, dzs, Craw, csraw = torch.empty(H, device=DEV), torch.empty(1, device=DEV), torch.empty(H, H, device=DEV), torch.empty(H, device=DEV)
        S._lib.check(L.sgs_gemm_tn_mask_gather(bits.data_ptr(), dz.data_ptr(), w2.data_ptr(), scale, codes.data_ptr(), N, sd.data_ptr(), K, H, H,
                                               C.data_ptr(), 2 * H, cs.data_ptr(), dzs.data_ptr(), Craw.data_ptr(), csraw.data_ptr(), ws.data_ptr(),
                                               ws.numel(), ops._stream()), "gather")
        torch.cuda.synchronize()
        return C, cs, dzs, Craw, csraw

    try:
        new = run(1)
        new2 = run(1)
        odd = run(1, 37)                      # a slab count that leaves ragged halves and an empty trailing one
        old = run(0)
    finally:
        L.sgs_gemm_tn_set_gather_variant(1, 0)
    for a, b in zip(new, new2):
        assert torch.equal(a, b)              # run-to-run deterministic
    bw = torch.arange(32, device=DEV, dtype=torch.int32)
    mask = ((bits.unsqueeze(2) >> bw) & 1).reshape(K, H).double()
    feat = (codes[sd[:, 0].long()] * codes[sd[:, 1].long()]).double()
    dv = mask * dz.double().unsqueeze(1)
    Traw = dv.t() @ feat
    csr = dv.sum(0)
    want = (Traw * (w2.double() * scale).unsqueeze(1), csr * w2.double() * scale, dz.double().sum().reshape(1), Traw, csr)
    for got in (new, odd, old):
        assert torch.all(got[0][:, H:] == 7.0)                       # the strided result leaves the other half of the rows alone
        for name, a, b in zip(("C", "colsum", "dzsum", "Craw", "colsum_raw"), (got[0][:, :H], got[1], got[2], got[3], got[4]), want):
            err = float((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-30))
            assert err < 2e-6, (name, err)

