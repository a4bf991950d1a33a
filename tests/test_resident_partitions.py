"""ResidentPartitions (device-side ClusterData / ClusterLoader counterpart): CPU test of the partitioning logic against a
straightforward per-partition restatement, and a GPU test that the resident batches train."""
import argparse

import pytest
import torch


def _graph(n=90, e=900, seed=0):
    import sgs_gnn_amd as S
    b = S.synthetic_graph(n, e, 7, 3, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    part = torch.randint(0, 5, (n,), generator=g)
    return S, b, part


def _check(S, b, part, rp, dev):
    n = b.x.shape[0]
    seen_edges = 0
    for p in range(5):
        nodes = torch.nonzero(part == p).squeeze(1)
        bt = rp[p]
        assert torch.equal(bt.node_ids.cpu(), nodes)
        assert torch.equal(bt.x.cpu(), b.x[nodes]) and torch.equal(bt.y.cpu(), b.y[nodes])
        assert torch.equal(bt.train_mask.cpu(), b.train_mask[nodes]) and torch.equal(bt.test_mask.cpu(), b.test_mask[nodes])
        loc = -torch.ones(n, dtype=torch.int64)
        loc[nodes] = torch.arange(nodes.numel())
        s, d = b.edge_index
        keep = (part[s] == p) & (part[d] == p)
        ref = torch.stack([loc[s[keep]], loc[d[keep]]])
        order = torch.argsort(ref[0] * n + ref[1])
        assert torch.equal(bt.edge_index.cpu(), ref[:, order])
        torch.testing.assert_close(bt.prob.cpu(), b.prob[keep][order], rtol=1e-5, atol=1e-9)     # global prior, sliced
        seen_edges += int(keep.sum())
    assert rp.dropped_edges == b.edge_index.shape[1] - seen_edges
    assert len(rp) == 5 and sorted(bt.part for bt in rp) == list(range(5))


def test_resident_partitions_cpu_logic():
    S, b, part = _graph()
    rp = S.ResidentPartitions(b.x, b.edge_index, b.y, b.train_mask, b.val_mask, b.test_mask, part, num_parts=5, device="cpu")
    _check(S, b, part, rp, "cpu")
    rl = S.ResidentPartitions(b.x, b.edge_index, b.y, b.train_mask, b.val_mask, b.test_mask, part, num_parts=5, device="cpu", prior="local")
    for bt in rl:
        torch.testing.assert_close(bt.prob, S.degree_prior(bt.edge_index, bt.x.shape[0]))
    a = [bt.part for bt in S.ResidentPartitions(b.x, b.edge_index, b.y, b.train_mask, b.val_mask, b.test_mask, part, 5, "cpu", shuffle=True, seed=3)]
    assert sorted(a) == list(range(5))


def test_partition_cache_round_trip(tmp_path):
    """save() -> load(): the tensor-only safetensors cache (data.py: "on-disk partition cache") reproduces every resident batch bit for
    bit, refuses a file with another partition count, and holds plain tensors + string metadata only."""
    from safetensors import safe_open
    S, b, part = _graph()
    rp = S.ResidentPartitions(b.x, b.edge_index, b.y, b.train_mask, b.val_mask, b.test_mask, part, num_parts=5, device="cpu")
    path = rp.save(str(tmp_path))
    assert path.endswith("sgs_partitions_5.safetensors")
    with safe_open(path, framework="pt") as f:
        assert f.metadata()["format"] == "sgs-partitions-v1"
        assert sorted(f.keys()) == sorted(["perm", "node_ptr", "edge_ptr", "x", "y", "train_mask", "val_mask", "test_mask", "edge_index", "prob"])
    rl = S.ResidentPartitions.load(str(tmp_path), 5, device="cpu")
    _check(S, b, part, rl, "cpu")
    for a, c in zip(rp, rl):
        for k in ("x", "edge_index", "y", "train_mask", "val_mask", "test_mask", "prob", "node_ids"):
            assert torch.equal(getattr(a, k), getattr(c, k)), k
        assert getattr(c, "edge_index").is_contiguous()
    assert rl.dropped_edges == rp.dropped_edges
    with pytest.raises((ValueError, FileNotFoundError)):
        S.ResidentPartitions.load(str(tmp_path), 4, device="cpu")


@pytest.mark.gpu
def test_resident_partitions_on_device_and_train():
    S, b, part = _graph(n=400, e=12000, seed=4)
    rp = S.ResidentPartitions(b.x, b.edge_index, b.y, b.train_mask, b.val_mask, b.test_mask, part, num_parts=5, device="cuda:0")
    _check(S, b, part, rp, "cuda:0")
    m = S.GNNModel(7, 16, 3, dropout_prob=0.2, edge_mlp_type="GCN").to("cuda:0")
    og = S.FusedAdam([p for n, p in m.named_parameters() if "gcn" in n], lr=1e-2)
    oe = S.FusedAdam([p for n, p in m.named_parameters() if "edge_prob_mlp" in n], lr=1e-2)
    args = argparse.Namespace(device="cuda:0", mode="learned", pipeline="hybrid", conditional=True, sparse_edge_mlp=True, t_init=0.7,
                              t_min=0.5, degree_bias_coef=0.3, reg1=True, reg2=True, regularizer1_coef=1.0, consist_reg_coef=0.5,
                              hybrid_checkpoint=False, sgs_hipgraph=True)
    for ep in range(4):
        loss, _, cond, tot = S.train(args, ep, 4, m, og, oe, None, torch.nn.CrossEntropyLoss(), rp, q=300)
        assert tot == 5 and loss == loss
    assert 1 <= m._sgs_stepgraphs.captures <= 4                    # one capture per slot, whatever the number of partitions
