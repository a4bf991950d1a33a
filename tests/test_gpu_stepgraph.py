"""GPU: opt-in HIP-graph replay of the partition step (stepgraph.py) against the eager path.

Replays draw their own noise (RNG epoch word), so the sampled step is checked by recomputing it eagerly
from the replay's own draws (dropout 0); unsampled partitions must match eager training bit for bit."""
import argparse
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _args(**kw):
    a = argparse.Namespace(device=DEV, mode="learned", pipeline="hybrid", edge_mlp_type="GCN", conditional=True,
                           sparse_edge_mlp=True, t_init=0.7, t_min=0.5, degree_bias_coef=0.3, reg1=True, reg2=True,
                           regularizer1_coef=1.0, consist_reg_coef=0.5, hybrid_checkpoint=False, drop_rate=0.0, lr=1e-2)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def _setup(S, p_drop, seed=3, capturable=False, hid=32):
    torch.manual_seed(seed)
    S.fix_seeds(seed)
    m = S.GNNModel(24, hid, 5, dropout_prob=p_drop, edge_mlp_type="GCN").to(DEV)
    kw = dict(capturable=True, fused=True) if capturable else {}
    og = torch.optim.Adam([p for n, p in m.named_parameters() if "gcn" in n], lr=1e-2, **kw)
    oe = torch.optim.Adam([p for n, p in m.named_parameters() if "edge_prob_mlp" in n], lr=1e-2, **kw)
    return m, og, oe


def _batches(S, sizes, n=120, seed=11):
    return [S.synthetic_graph(n, E, 24, 5, seed=seed + i, device=DEV) for i, E in enumerate(sizes)]


@pytest.mark.parametrize("capturable", [False, True])
def test_unsampled_partitions_replay_equals_eager_bitwise(capturable):
    """capturable=True: the optimiser steps are recorded at the end of the backward graphs (no eager launch per step)."""
    import sgs_gnn_amd as S
    crit = torch.nn.CrossEntropyLoss()
    # every partition below q: no draws, dropout 0 -> deterministic.  Sizes: a replayed step runs the kernels chosen for the slot's
    # CAPACITY, an eager step those chosen for the partition's own size (sgs_spmm_csr: a workgroup per row from 16 entries per
    # row on); all of these lie on the same side of that switch, so the arithmetic is the same, in the same order.
    bs = _batches(S, [2500, 3300, 2100])
    q = 5000
    m1, og1, oe1 = _setup(S, 0.0, capturable=capturable)
    m2, og2, oe2 = _setup(S, 0.0, capturable=capturable)
    m2.load_state_dict(copy.deepcopy(m1.state_dict()))
    r1, r2 = [], []
    for ep in range(4):                                # every step of the graph-mode run is a replay, the very first included
        r1.append(S.train(_args(), ep, 10, m1, og1, oe1, None, crit, bs, q=q))
        r2.append(S.train(_args(sgs_hipgraph=True), ep, 10, m2, og2, oe2, None, crit, bs, q=q))
    assert r1 == r2
    for (n1, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        assert torch.equal(p1, p2), n1
    sg = m2._sgs_stepgraphs
    assert sg.captures == 2 and len(sg.slots[False]) == 2 and not sg.slots[True]      # ONE capture per slot serves all partitions
    assert (sg.optimizers is not None) == capturable
    # the RNG epoch word is registered only while a graph-mode train() runs
    assert S.ops._rng_epoch is None


def _cmp(a, b, rtol, atol, name="", rel_max=None):
    """allclose(a, b, rtol, atol); with `rel_max`: max |a - b| <= rel_max * max |b| (the at-size tests' bound: a gradient tensor
    spans orders of magnitude, and its small entries carry the summation-order noise of its large ones)."""
    if rel_max is None:
        assert torch.allclose(a, b, rtol=rtol, atol=atol), (name, float((a - b).abs().max()))
    else:
        err, ref = float((a - b).abs().max()), float(b.abs().max())
        assert err <= rel_max * max(ref, 1e-30), (name, err, ref)


def _check_sampled_replay(S, m, a, crit, b, q, pipeline, k, cnt, gl, ll, gr, lr_, rel_max=None):
    """Recompute one replayed sampled step eagerly from the replay's own draws (`k`: the slot's kept views, already cut to the
    partition's E edges / N nodes) and compare outputs, both losses and both branches' gradients."""
    from sgs_gnn_amd.training import _ce, learned_loss, SampledForward
    ops = S.ops
    N = b.x.shape[0]
    params = list(m.parameters())
    assert k["eid"].numel() == q and bool((k["eid"][1:] > k["eid"][:-1]).all())
    assert int(k["eid"][-1]) < b.edge_index.shape[1]
    assert torch.equal(k["sampled_edge_index"], b.edge_index[:, k["eid"]])
    for p in params:
        p.grad = None
    sc = m.edge_prob_mlp
    st = SampledForward()
    st.rsei, st.sampled_edge_index = k["rsei"], k["sampled_edge_index"]
    if pipeline == "two_pass":
        with torch.no_grad():
            pf = sc(b.x, b.edge_index, k["rsei"]).squeeze()
    else:
        pf = sc(b.x, b.edge_index, k["rsei"]).squeeze()
    assert torch.allclose(pf.detach(), k["edge_probs_full"], rtol=1e-5, atol=1e-6)
    if pipeline == "hybrid":
        act = getattr(sc, "last_active", None)
        if act is not None:
            act.set(k["eid"], ops.get_graph(k["sampled_edge_index"], N))
        w = pf.index_select(0, k["eid"])
    elif pipeline == "two_pass":
        w = sc(b.x, k["sampled_edge_index"]).squeeze()
    else:
        w = None
    if w is not None:
        st.edge_probs_for_loss = w
        assert torch.allclose(w.detach(), k["w"], rtol=1e-5, atol=1e-6)
        st.learned_out = m(b, k["sampled_edge_index"], w)
        assert torch.allclose(st.learned_out.detach(), k["learned_out"], rtol=1e-4, atol=1e-5)
        loss = learned_loss(a, crit, st, b)
        loss.backward()
        assert torch.allclose(loss.detach(), ll, rtol=1e-5, atol=1e-6)
        for i, p in enumerate(params):
            if p.grad is None:
                assert i not in gl
            else:
                _cmp(p.grad, gl[i], 2e-4, 2e-6, ("learned grad", i), rel_max)
    # random branch
    for p in params:
        p.grad = None
    ro = m(b, k["rsei"])
    assert torch.allclose(ro.detach(), k["random_out"], rtol=1e-4, atol=1e-5)
    lr2 = _ce(crit, ro, b)
    lr2.backward()
    assert torch.allclose(lr2.detach(), lr_, rtol=1e-5, atol=1e-6)
    for i, p in enumerate(params):
        if p.grad is None:
            assert i not in gr
        else:
            _cmp(p.grad, gr[i], 2e-4, 2e-6, ("random grad", i), rel_max)
    want = [int(x) for x in ops.masked_correct(k["learned_out"], b.y, b.train_mask).tolist()]
    assert cnt[0:2] == want


def _kept(c, b):
    """The slot's kept views cut to the staged partition's own sizes (slot buffers are capacity-sized)."""
    E, N = b.edge_index.shape[1], b.x.shape[0]
    k = {n: (None if t is None else t.clone()) for n, t in c.keep.items()}
    k["edge_probs_full"] = k["edge_probs_full"][:E]
    for n in ("learned_out", "random_out"):
        if k[n] is not None:
            k[n] = k[n][:N]
    return k


@pytest.mark.parametrize("pipeline", ["hybrid", "straight_through", "two_pass"])
def test_sampled_replay_matches_eager_recomputation_from_its_own_draws(pipeline):
    import sgs_gnn_amd as S
    from sgs_gnn_amd.stepgraph import StepGraphs
    crit = torch.nn.CrossEntropyLoss()
    b = _batches(S, [4000])[0]
    q = 800
    m, og, oe = _setup(S, 0.0)
    a = _args(pipeline=pipeline)
    sg = StepGraphs.attach(m, pipeline, a, crit, q, False, loader=[b])
    sg.debug_keep = True
    E = b.edge_index.shape[1]
    try:
        sg.step(b, 0)                                  # stage into a slot, warm up, capture, first replay
        for p in m.parameters():
            p.grad = None
        c = next(s_ for s_ in sg.slots[True] if s_.live is b)
        assert c.ecap >= E and c.ecap % 2048 == 0 and int(c.dims[0]) == E
        seen = []
        for it in range(3):
            sg.replay_g1(c)
            k = _kept(c, b)
            cnt = c.cbuf.tolist()
            seen.append(k["eid"].clone())
            c.g2l.replay()
            gl = {i: g.clone() for i, g in c.grads_l.items()}
            ll = c.loss_l.clone()
            c.g2r.replay()
            gr = {i: g.clone() for i, g in c.grads_r.items()}
            lr_ = c.loss_r.clone()
            torch.cuda.synchronize()
            _check_sampled_replay(S, m, a, crit, b, q, pipeline, k, cnt, gl, ll, gr, lr_)
        # every replay drew a different edge set; resetting the epoch word reproduces a replay exactly
        assert not torch.equal(seen[0], seen[1]) and not torch.equal(seen[1], seen[2])
        e = int(sg.epoch_word.item())
        sg.epoch_word.fill_(e - 2)                     # each iteration replayed G2L and G2R: two epoch ticks after the last G1
        sg.replay_g1(c)
        torch.cuda.synchronize()
        assert torch.equal(c.keep["eid"], seen[2])
    finally:
        sg.release()


@pytest.mark.parametrize("pipeline,hid,shapes,q", [
    ("hybrid", 32, [(150, 6100), (90, 2600), (120, 4000), (110, 900), (140, 1500), (100, 700)], 1000),
    ("straight_through", 32, [(150, 6100), (90, 2600), (120, 4000), (110, 900), (140, 1500), (100, 700)], 1000),
    # H = 128 and > 65 536 candidate edges: the PAIRED bf16x6 scorer forward inside the capture, its live number of canonical
    # edges read from the slot's second dims word
    ("hybrid", 128, [(420, 90_000), (380, 70_000), (400, 80_000), (300, 9_000)], 15_000)])
def test_one_capture_serves_partitions_of_different_sizes(pipeline, hid, shapes, q):
    """The step is captured ONCE per slot, over static buffers sized for the largest partition; a partition is handed over by one
    staging launch and the kernels over the candidate edges read the live edge count from the slot.  Partitions with different
    numbers of nodes (padded with isolated, unlabelled nodes) and edges -- visited in an order that leaves the leftovers of a
    LARGER partition behind in the slot -- each reproduce their eager recomputation; nothing is captured after the first visits."""
    import sgs_gnn_amd as S
    from sgs_gnn_amd.stepgraph import StepGraphs
    from sgs_gnn_amd.training import _ce
    crit = torch.nn.CrossEntropyLoss()
    bs = [S.synthetic_graph(n, E, 24, 5, seed=40 + i, device=DEV) for i, (n, E) in enumerate(shapes)]
    m, og, oe = _setup(S, 0.0, hid=hid)
    a = _args(pipeline=pipeline)
    sg = StepGraphs.attach(m, pipeline, a, crit, q, False, loader=bs)
    sg.debug_keep = True
    params = list(m.parameters())
    try:
        for rnd in range(2):
            for b in bs:
                E, N = b.edge_index.shape[1], b.x.shape[0]
                h = sg.forward(b)
                c = h.c
                assert c.live is b and int(c.dims[0]) == E and c.npad == max(n_ for n_, _ in shapes)
                if h.sampled and hid == 128:
                    assert c.canon is not None and E // 2 <= int(c.dims[1]) < E          # the paired forward ran, half the edges canonical
                if h.sampled:
                    cnt = h.gate_counts()
                    k = _kept(c, b)
                    c.g2l.replay()
                    gl = {i: g.clone() for i, g in c.grads_l.items()}
                    ll = c.loss_l.clone()
                    c.g2r.replay()
                    gr = {i: g.clone() for i, g in c.grads_r.items()}
                    lr_ = c.loss_r.clone()
                    sg.host_epoch += 2
                    torch.cuda.synchronize()
                    _check_sampled_replay(S, m, a, crit, b, q, pipeline, k, cnt, gl, ll, gr, lr_)
                else:
                    loss = h.backward(None).clone()
                    got = {i: g.clone() for i, g in c.grads.items()}
                    torch.cuda.synchronize()
                    for p in params:
                        p.grad = None
                    ref = _ce(crit, m(b, b.edge_index), b)
                    ref.backward()
                    assert torch.allclose(ref.detach(), loss, rtol=1e-5, atol=1e-6)
                    for i, p in enumerate(params):
                        if p.grad is None:
                            assert i not in got
                        else:
                            assert torch.allclose(p.grad, got[i], rtol=2e-4, atol=2e-6), i
                for p in params:
                    p.grad = None
        # two slots per kind, captured on their first use, nothing afterwards (the third list has ONE unsampled partition, which
        # stays live in its slot: the second unsampled slot is never needed)
        assert sg.captures == (4 if hid == 32 else 3)
        if hid == 128:
            # and in eager mode the same model takes the paired forward as well (the recomputation above went through it)
            assert S.ops.get_pairs(bs[0].edge_index, bs[0].x.shape[0]) is not None
    finally:
        sg.release()


@pytest.mark.parametrize("capturable", [False, True])
def test_graph_mode_training_with_dropout_runs_and_learns(capturable):
    import sgs_gnn_amd as S
    crit = torch.nn.CrossEntropyLoss()
    bs = _batches(S, [5000, 900, 4000], n=150)
    q = 1000
    m, og, oe = _setup(S, 0.3, capturable=capturable)
    before = {n: p.detach().clone() for n, p in m.named_parameters()}

    def eval_ce():
        m.eval()
        with torch.no_grad():
            v = sum(float(torch.nn.functional.cross_entropy(m(b, b.edge_index)[b.train_mask], b.y[b.train_mask])) for b in bs)
        m.train()
        return v

    ce0 = eval_ce()
    a = _args(sgs_hipgraph=True)
    losses, conds = [], 0
    for ep in range(12):
        loss, _, cond, tot = S.train(a, ep, 12, m, og, oe, None, crit, bs, q=q)
        assert tot == 3 and 0 <= cond <= 2
        if capturable:
            # optimiser state exists before any step is captured (state created inside a capture would be reset by each replay)
            assert all(len(o.state[p]) > 0 for o in (og, oe) for grp in o.param_groups for p in grp["params"])
        conds += cond
        losses.append(loss)
    assert all(torch.isfinite(torch.tensor(losses)))
    assert eval_ce() < ce0                     # full-graph, eval-mode CE on the train rows went down
    for n, p in m.named_parameters():
        assert torch.isfinite(p).all()
    moved = [n for n, p in m.named_parameters() if not torch.equal(p, before[n])]
    assert any("gcn1" in n for n in moved)
    if conds:
        assert any("fc1" in n for n in moved)
    # eager evaluation afterwards is unaffected by graph mode
    args_e = argparse.Namespace(degree_bias_coef=0.3, num_samples_eval=2)
    f1 = S.evaluate(args_e, m, bs, DEV, q=q, mode="learned")
    assert all(0.0 <= v <= 1.0 for v in f1)


def test_graph_mode_with_gat_model_straight_through():
    """Config 4 (GAT encoder, straight-through pipeline) through the replayed step: attention dropout draws from the RNG
    epoch word too; parameter names follow main.py:100-109 ('GAT' goes to the GNN optimiser)."""
    import sgs_gnn_amd as S
    torch.manual_seed(5)
    S.fix_seeds(5)
    crit = torch.nn.CrossEntropyLoss()
    bs = _batches(S, [5000, 900, 4000], n=150)
    q = 1000
    m = S.GATModel(24, 32, 5, dropout_prob=0.3, edge_mlp_type="GCN").to(DEV)
    og = S.FusedAdam([p for n, p in m.named_parameters() if "GAT" in n or "gcn" in n], lr=1e-2)
    oe = S.FusedAdam([p for n, p in m.named_parameters() if "edge_prob_mlp" in n], lr=1e-2)
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    a = _args(sgs_hipgraph=True, pipeline="straight_through")
    for ep in range(6):
        loss, _, cond, tot = S.train(a, ep, 6, m, og, oe, None, crit, bs, q=q)
        assert tot == 3 and loss == loss
    for n, p in m.named_parameters():
        assert torch.isfinite(p).all(), n
    assert any(not torch.equal(p, before[n]) for n, p in m.named_parameters() if "GAT" in n)
    assert m._sgs_stepgraphs.captures <= 4


def test_prefix_prefetch_on_a_second_stream_changes_nothing(monkeypatch):
    """The parameter-independent head of a sampled step (prior draw, CSR of the random graph, unit norm) is its own graph G0;
    with the following batch named it is replayed a step ahead on a second stream.  Same kernels, same epoch values: the
    run must be bit-identical to the one that replays G0 in line, with dropout on."""
    import sgs_gnn_amd as S
    from sgs_gnn_amd import stepgraph
    crit = torch.nn.CrossEntropyLoss()
    bs = _batches(S, [5000, 900, 4000, 6000, 4500], n=150)
    q = 1000
    results = []
    for prefetch in (True, False):
        monkeypatch.setattr(stepgraph, "_PREFETCH", prefetch)
        torch.manual_seed(3)
        S.fix_seeds(3)
        m = S.GNNModel(24, 32, 5, dropout_prob=0.3, edge_mlp_type="GCN").to(DEV)
        og = S.FusedAdam([p for n, p in m.named_parameters() if "gcn" in n], lr=1e-2)              # main.py:100 (overlap kept)
        oe = S.FusedAdam([p for n, p in m.named_parameters() if "edge_prob_mlp" in n], lr=1e-2)
        a = _args(sgs_hipgraph=True)
        rets = [S.train(a, ep, 8, m, og, oe, None, crit, bs, q=q) for ep in range(8)]
        sg = m._sgs_stepgraphs
        assert sg.slots[True] and all(c.g0 is not None for c in sg.slots[True])
        # the host's mirror of the epoch word (what a prefetched G0 is given) agrees with the device
        assert int(sg.epoch_word.item()) == sg.host_epoch
        results.append((rets, {n: p.detach().clone() for n, p in m.named_parameters()}))
    (r_on, p_on), (r_off, p_off) = results
    assert r_on == r_off
    for n in p_on:
        assert torch.equal(p_on[n], p_off[n]), n


# --------------------------------------------------------------------------------------------------------------------------------
# The BENCHMARKED step at the benchmarked shape (bench.py's S3 stream: n = 1 013, F = 602, H = 256, C = 41, q = 100 000, dropout 0.3,
# in-graph FusedAdam): HIP-graph replay with a live E below the slot's capacity, the paired mask-keeping scorer forward, the mask-form
# backward at >= 65 536 active rows (sgs_edge_score_bwd_prep, sgs_gemm_tn_mask, sgs_endpoint_reduce_pair_bits, dw2_from_parts) and
# both Adam steps inside the backward graph.
def _adam_rule(p, g, st, lr, betas=(0.9, 0.999), eps=1e-8):
    """torch.optim.Adam's update in fp64: returns (new parameter, new state)."""
    import math
    b1, b2 = betas
    t = float(st["step"]) + 1.0
    m = st["exp_avg"].double()
    m = m + (g.double() - m) * (1.0 - b1)
    v = b2 * st["exp_avg_sq"].double() + (1.0 - b2) * g.double() ** 2
    bc1, bc2 = 1.0 - b1 ** t, 1.0 - b2 ** t
    denom = v.sqrt() / math.sqrt(bc2) + eps
    return p.double() - (lr / bc1) * (m / denom), dict(step=t, exp_avg=m, exp_avg_sq=v)


def _opt_snapshot(opts):
    return [{p: {k_: v.clone() for k_, v in o.state[p].items()} for grp in o.param_groups for p in grp["params"]} for o in opts]


def _opt_restore(opts, snap):
    for o, sn in zip(opts, snap):
        for p, st in sn.items():
            for k_, v in st.items():
                o.state[p][k_].copy_(v)


def test_replayed_step_at_the_benchmarked_shape_matches_its_eager_recomputation():
    import sgs_gnn_amd as S
    from sgs_gnn_amd.model import _DropoutClock
    from sgs_gnn_amd.stepgraph import StepGraphs
    n, F_, H, C, q, lr = 1013, 602, 256, 41, 100_000, 1e-3
    crit = torch.nn.CrossEntropyLoss()
    # two larger partitions first (one per slot of the ping-pong), then two smaller ones: each of those is staged into a slot
    # that still holds the leftovers of a LARGER partition past its live edge count
    bs = [S.synthetic_graph(n, E, F_, C, seed=70 + i, device=DEV) for i, E in enumerate([350_000, 300_000, 210_000, 120_000])]
    torch.manual_seed(4)
    S.fix_seeds(4)
    m = S.GNNModel(F_, H, C, dropout_prob=0.3, edge_mlp_type="GCN").to(DEV)
    og = S.FusedAdam([p for nme, p in m.named_parameters() if "gcn" in nme], lr=lr)                # main.py:100 (overlap with the scorer's encoder kept)
    oe = S.FusedAdam([p for nme, p in m.named_parameters() if "edge_prob_mlp" in nme], lr=lr)      # main.py:122
    a = _args(pipeline="hybrid", drop_rate=0.3, lr=lr)
    sg = StepGraphs.attach(m, "hybrid", a, crit, q, False, optimizers=(oe, og), loader=bs)
    sg.debug_keep = True
    params = list(m.parameters())
    names = [nme for nme, _ in m.named_parameters()]
    assert sg.optimizers is not None and sg.pairs_ok
    hosted = {}
    try:
        for step, b in enumerate(bs):
            E = b.edge_index.shape[1]
            P0 = [p.detach().clone() for p in params]
            S0 = _opt_snapshot((oe, og))
            h = sg.forward(b)
            c = h.c
            hosted.setdefault(id(c), []).append(E)
            assert h.sampled and c.live is b and int(c.dims[0]) == E and c.ecap >= 350_000
            assert c.canon is not None and E // 2 <= int(c.dims[1]) < E                 # the paired forward ran over the canonical half
            cnt = h.gate_counts()
            e0 = int(sg.epoch_word.item())                                               # the epoch every kernel of this step folded into its seeds
            k = _kept(c, b)
            # ---- learned branch: backward + both Adam steps inside G2L
            c.g2l.replay()
            torch.cuda.synchronize()
            gl = {i: g.clone() for i, g in c.grads_l.items()}
            ll = c.loss_l.clone()
            P1 = [p.detach().clone() for p in params]
            S1 = _opt_snapshot((oe, og))
            assert len(gl) == 12                                                         # every tensor of scorer + GNN has a gradient
            # post-Adam parameters = Adam's rule on the replay's own gradients: optimizer_edge_prob first, then optimizer_gnn
            # (training_hybrid.py:136-137; the two overlap on edge_prob_mlp.gcn*, which therefore step twice)
            cur = {i: P0[i] for i in range(len(params))}
            for o, sn in zip((oe, og), S0):
                for grp in o.param_groups:
                    for p in grp["params"]:
                        i = next(j for j, pp in enumerate(params) if pp is p)
                        new, _ = _adam_rule(cur[i], gl[i], sn[p], lr)
                        cur[i] = new.float()
            for i in range(len(params)):
                torch.testing.assert_close(P1[i], cur[i], rtol=1e-5, atol=1e-7, msg=lambda s_, i=i: f"post-Adam {names[i]}: {s_}")
                assert not torch.equal(P1[i], P0[i]), names[i]
            # ---- random branch from the same forward: parameters and optimiser state put back first (G2R's dX products read W)
            for p, v in zip(params, P0):
                p.data.copy_(v)
            _opt_restore((oe, og), S0)
            c.g2r.replay()
            torch.cuda.synchronize()
            gr = {i: g.clone() for i, g in c.grads_r.items()}
            lr_ = c.loss_r.clone()
            assert sorted(names[i] for i in gr) == ["gcn1.bias", "gcn1.lin.weight", "gcn2.bias", "gcn2.lin.weight"]
            for i, g in gr.items():
                new, _ = _adam_rule(P0[i], g, S0[1][params[i]], lr)
                torch.testing.assert_close(params[i].detach(), new.float(), rtol=1e-5, atol=1e-7, msg=lambda s_, i=i: f"post-Adam (random) {names[i]}: {s_}")
            # ---- eager recomputation from the replay's own draws: the parameters of before the step, the same dropout seeds
            # (frozen at capture) and the same RNG epoch
            for p, v in zip(params, P0):
                p.data.copy_(v)
            _opt_restore((oe, og), S0)
            sg.epoch_word.fill_(e0)
            tick_now = _DropoutClock.tick
            _DropoutClock.tick = sg.seed_state[True][1]
            S.ops.new_memo_scope()
            try:
                _check_sampled_replay(S, m, a, crit, b, q, "hybrid", k, cnt, gl, ll, gr, lr_, rel_max=2e-4)
            finally:
                _DropoutClock.tick = tick_now
            # ---- carry on as training would have: the learned branch's update, two epoch ticks (G2L and G2R were both replayed)
            for p, v in zip(params, P1):
                p.data.copy_(v)
                p.grad = None
            _opt_restore((oe, og), S1)
            sg.epoch_word.fill_(e0 + 2)
            sg.host_epoch += 2
            S.ops.drop_memos(m)
        assert sg.captures == 2                                                          # one capture per slot, nothing after the first visits
        for sizes in hosted.values():
            assert len(sizes) == 2 and sizes[1] < sizes[0]                               # the smaller partition came AFTER the larger one in that slot
    finally:
        sg.release()


def test_workspace_guard_and_first_visit_prefetch_of_an_unscannable_loader():
    """(1) The scratch-arena guard: a slot is owned by the stream that last took scratch from it; another stream asking for it raises
    unless the hand-over was declared (the fault of round 2's first bench -- a CSR build on the prefetch stream sharing arena 0 with
    the replayed step's sampler -- now raises instead of corrupting memory).  (2) The path that used to do it: a loader that
    reserve() cannot scan (a generator), so a partition's FIRST hand-over happens in _prefetch(): its CSR / mates are built on the
    main stream, only the copy runs on the prefetch stream.  With the guard on, an epoch over such a loader must pass and train."""
    import sgs_gnn_amd as S
    ops = S.ops
    assert ops._ws_guard, "tests/conftest.py switches the guard on before the package is imported"
    dev = torch.device(DEV)
    ops.workspace(1024, dev)                                   # owned by the current (default) stream
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with pytest.raises(RuntimeError, match="hand-over"):
            ops.workspace(1024, dev)
        with ops.workspace_slot(7):                            # its own arena: fine
            ops.workspace(1024, dev)
        side.wait_stream(torch.cuda.default_stream())
        ops.workspace_handover(dev)                            # declared: the side stream owns the arenas now
        ops.workspace(1024, dev)
    with pytest.raises(RuntimeError, match="hand-over"):
        ops.workspace(1024, dev)                               # ... and the default stream has to take them back the same way
    torch.cuda.current_stream().wait_stream(side)
    ops.workspace_handover(dev)
    ops.workspace(1024, dev)

    crit = torch.nn.CrossEntropyLoss()
    q = 1000
    m, og, oe = _setup(S, 0.3)
    og = S.FusedAdam([p for n, p in m.named_parameters() if "gcn" in n], lr=1e-2)
    oe = S.FusedAdam([p for n, p in m.named_parameters() if "edge_prob_mlp" in n], lr=1e-2)
    a = _args(sgs_hipgraph=True)
    sizes = [5000, 900, 4000, 6000, 800, 4500, 5200]
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    for ep in range(2):
        # fresh batch OBJECTS every epoch, produced lazily: nothing is cached on them ahead of their first hand-over
        loader = (S.synthetic_graph(150, E, 24, 5, seed=200 + 10 * ep + i, device=DEV) for i, E in enumerate(sizes))

        class Gen:
            def __iter__(self_):
                return loader

            def __len__(self_):
                return len(sizes)
        loss, _, cond, tot = S.train(a, ep, 2, m, og, oe, None, crit, Gen(), q=q)
        assert tot == len(sizes) and loss == loss
    sg = m._sgs_stepgraphs
    assert sg.last_pre is not None                             # hand-overs did go through the prefetch stream
    assert any(not torch.equal(p, before[n]) for n, p in m.named_parameters())


def test_profiler_segments_are_the_references_and_emit_roctx_ranges():
    """model.gpu_profiler / model.edge_prob_mlp.gpu_profiler (main.py:116-119): begin / end around the reference's four segments
    (`edge_mlp_pre`, `edge_score`, `gnn_forward`, `backward`), each also a rocTX range; summary keys as utils.py:51-74."""
    import sgs_gnn_amd as S
    from sgs_gnn_amd.utils import SEGMENTS, _RocTx
    assert _RocTx.lib() is not None
    crit = torch.nn.CrossEntropyLoss()
    b = _batches(S, [4000])[0]
    m, og, oe = _setup(S, 0.3)
    prof = S.GpuMemoryProfiler(enabled=True, device=DEV)
    assert prof.enabled
    m.gpu_profiler = prof
    m.edge_prob_mlp.gpu_profiler = prof
    prof.start_epoch(0)
    S.train(_args(), 0, 1, m, og, oe, None, crit, [b], q=800)
    summ = prof.summarize_epoch(0)
    prof.end_epoch()
    assert set(summ) == set(SEGMENTS)
    assert summ["edge_mlp_pre"]["calls"] == 1 and summ["edge_score"]["calls"] == 1 and summ["backward"]["calls"] == 1
    assert summ["gnn_forward"]["calls"] == 2                   # learned + random forward (training_hybrid.py:88,93)
    for d in summ.values():
        assert {"max_peak_inc_bytes", "max_peak_inc_mb", "mean_peak_inc_mb", "max_alloc_after_mb", "max_alloc_inc_mb", "calls"} <= set(d)
    assert summ["edge_score"]["max_alloc_after_bytes"] > 0     # (peak increases are relative to the process-wide peak, as in the reference)
