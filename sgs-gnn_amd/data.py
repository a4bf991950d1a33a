"""Batch duck-type of the reference's loaders (PyG `Data` as ClusterLoader yields it: .x, .edge_index,
.y, .train_mask/.val_mask/.test_mask, .prob, .to(device)) and seeded synthetic look-alikes of the
benchmark graphs: real datasets cannot be fetched offline."""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


class Batch:
    def __init__(self, **kw):
        self.__dict__.update(kw)

    @property
    def num_nodes(self):
        return self.x.shape[0]

    def to(self, device):
        dev = torch.device(device)
        if self.x.device == dev or (dev.type == "cuda" and self.x.is_cuda and dev.index in (None, self.x.device.index)):
            return self
        out = Batch(**{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in self.__dict__.items() if not k.startswith("_sgs")})
        return out


def degree_prior(edge_index: torch.Tensor, num_nodes: int) -> torch.Tensor:
    """`data.prob` of datasets.py:141-156 (add_degree) for a row-sorted edge_index:
    softmax_e( E^-1/2 / (colcount[row_e] + rowcount[col_e] + 1e-10) )."""
    row, col = edge_index[0], edge_index[1]
    E = edge_index.shape[1]
    rowcount = torch.bincount(row, minlength=num_nodes).to(torch.float32)
    colcount = torch.bincount(col, minlength=num_nodes).to(torch.float32)
    prob = 1.0 / ((colcount[row] + rowcount[col]) + 1e-10)
    return F.softmax(prob * E ** -0.5, dim=0)


def synthetic_graph(n: int, n_edges_target: int, nfeat: int, ncls: int, seed: int, train_frac: float = 0.66,
                    power: float = 0.8, device="cpu") -> Batch:
    """Undirected, loop-free, coalesced, row-sorted degree-corrected random graph with power-law
    expected degrees (both directions stored), N(0,1) features with a weak class signal, uniform
    labels, Bernoulli masks and the reference's degree prior."""
    dev = torch.device(device)
    g = torch.Generator(device=dev).manual_seed(seed)
    kw = dict(generator=g, device=dev)
    wts = (torch.arange(1, n + 1, dtype=torch.float32, device=dev) ** (-power))
    wts = wts[torch.randperm(n, **kw)]
    m = n_edges_target // 2
    max_pairs = n * (n - 1) // 2
    m = min(m, int(max_pairs * 0.9))
    keys = torch.zeros(0, dtype=torch.int64, device=dev)
    while keys.numel() < m:                       # draw endpoint pairs ~ w_i w_j, dedupe, repeat
        need = int((m - keys.numel()) * 1.3) + 16
        a = torch.multinomial(wts, need, replacement=True, generator=g)
        b = torch.multinomial(wts, need, replacement=True, generator=g)
        a, b = a.to(torch.int64), b.to(torch.int64)
        ok = a != b
        lo, hi = torch.minimum(a[ok], b[ok]), torch.maximum(a[ok], b[ok])
        keys = torch.unique(torch.cat([keys, lo * n + hi]))
    keys = keys[torch.randperm(keys.numel(), **kw)[:m]]
    lo, hi = keys // n, keys % n
    both = torch.unique(torch.cat([lo * n + hi, hi * n + lo]))       # sorted -> row-sorted, coalesced
    ei = torch.stack([both // n, both % n])
    y = torch.randint(0, ncls, (n,), **kw)
    x = torch.randn(n, nfeat, **kw)
    x[torch.arange(n, device=dev), y % nfeat] += 1.5
    r = torch.rand(n, **kw)
    tm = r < train_frac
    vm = (r >= train_frac) & (r < train_frac + (1 - train_frac) * 0.3)
    return Batch(x=x, edge_index=ei, y=y, train_mask=tm, val_mask=vm, test_mask=~(tm | vm), prob=degree_prior(ei, n))


def reddit_partition_stream(num_parts: int = 230, seed: int = 42, nfeat: int = 602, ncls: int = 41, n: int = 1013,
                            e_lo: int = 60_000, e_hi: int = 500_000, frac_above_q: float = 0.52, q: int = 100_000,
                            device="cpu", only=None):
    """S3 of SURVEY.md section 8d: a Reddit-like METIS partition stream -- `num_parts` batches of
    ~1013 nodes whose intra-partition edge counts mirror the reference run (119 of 230 partitions
    exceed q = 100 000; logs/pipeline_hybrid.log:8).  `only` (a set of indices): build just those partitions, None elsewhere."""
    sizes = reddit_partition_sizes(num_parts, seed, e_lo, e_hi, frac_above_q, q)
    return [synthetic_graph(n, E, nfeat, ncls, seed * 1000 + i, device=device) if only is None or i in only else None
            for i, E in enumerate(sizes)]


def reddit_partition_sizes(num_parts: int = 230, seed: int = 42, e_lo: int = 60_000, e_hi: int = 500_000, frac_above_q: float = 0.52,
                           q: int = 100_000):
    """Edge counts of `reddit_partition_stream`'s partitions (same seed -> same list), without building them."""
    g = torch.Generator().manual_seed(seed)
    sizes = []
    for i in range(num_parts):
        # deterministic interleave: every prefix of the stream has ~frac_above_q of its partitions above q
        above = int((i + 1) * frac_above_q) > int(i * frac_above_q)
        lo, hi = (int(q * 1.02), e_hi) if above else (e_lo, int(q * 0.98))
        sizes.append(int(lo + (hi - lo) * float(torch.rand(1, generator=g))))
    return sizes


class ResidentPartitions:
    """All partitions of one graph resident in HBM: the device-side counterpart of the reference's
    `ClusterData(data, num_parts)` + `ClusterLoader(cluster_data, batch_size=1, shuffle=True)` (main.py:63-65; SURVEY.md
    section 8f item 2).  The reference re-slices every batch on the CPU and copies it to the GPU each time it is visited;
    here the partitioning is applied ONCE on the device (METIS itself stays a host preprocessing step: pass its node ->
    partition vector as `part_id`) and iteration yields `Batch` objects whose tensors never move again -- which is also
    what the HIP-graph replay of the step needs (captures are keyed on the batch tensors' addresses).

    Semantics follow ClusterData: partition p holds the nodes with part_id == p (in ascending original id), the edges with
    BOTH endpoints in p (relabelled, row-sorted), the node attributes x / y / masks, and the edge attribute `prob` sliced
    from the full graph's prior (`prior="global"`, what the reference does: add_degree runs before partitioning,
    datasets.py:141-156) or recomputed on the partition (`prior="local"`)."""

    def __init__(self, x, edge_index, y, train_mask, val_mask, test_mask, part_id, num_parts=None, device="cuda:0", prior="global",
                 prob=None, shuffle=False, seed=0):
        dev = torch.device(device)
        part_id = part_id.to(dev).to(torch.int64)
        N = x.shape[0]
        if part_id.numel() != N:
            raise ValueError("part_id must have one entry per node")
        P = int(num_parts) if num_parts is not None else int(part_id.max()) + 1
        ei = edge_index.to(dev)
        E = ei.shape[1]
        if prob is None and prior == "global":
            if dev.type == "cuda":
                from . import ops
                # the device op wants a row-sorted edge list (CSR build); sort a copy and scatter the prior back
                order = torch.argsort(ei[0] * N + ei[1])
                p_sorted = ops.degree_prior(ei[:, order].contiguous(), N)
                prob = torch.empty_like(p_sorted)
                prob[order] = p_sorted
            else:
                prob = degree_prior(ei, N)
        elif prob is not None:
            prob = prob.to(dev)
        # nodes grouped by partition (stable: ascending original id inside a partition)
        perm = torch.argsort(part_id, stable=True)
        counts = torch.bincount(part_id, minlength=P)
        nptr = torch.zeros(P + 1, dtype=torch.int64, device=dev)
        nptr[1:] = torch.cumsum(counts, 0)
        new_id = torch.empty(N, dtype=torch.int64, device=dev)
        new_id[perm] = torch.arange(N, device=dev)
        # intra-partition edges, relabelled, sorted by (partition, new src, new dst)
        ps, pd = part_id[ei[0]], part_id[ei[1]]
        keep = ps == pd
        src, dst, pe = new_id[ei[0][keep]], new_id[ei[1][keep]], ps[keep]
        order = torch.argsort(src * N + dst)            # new ids are grouped by partition, so this sorts by (p, src, dst)
        src, dst, pe = src[order], dst[order], pe[order]
        eprob = prob[keep][order] if prob is not None else None
        ecounts = torch.bincount(pe, minlength=P)
        eptr = torch.zeros(P + 1, dtype=torch.int64, device=dev)
        eptr[1:] = torch.cumsum(ecounts, 0)
        xs, ys = x.to(dev)[perm], y.to(dev)[perm]
        tm, vm, sm = train_mask.to(dev)[perm], val_mask.to(dev)[perm], test_mask.to(dev)[perm]
        nptr_h, eptr_h = nptr.tolist(), eptr.tolist()    # one read-back at construction time
        self.num_parts, self.perm, self.node_ptr, self.edge_ptr = P, perm, nptr, eptr
        self.dropped_edges = int(E - int(keep.sum()))    # inter-partition edges (ClusterLoader with batch_size=1 drops them too)
        self.batches = []
        for p in range(P):
            a, b, ea, eb = nptr_h[p], nptr_h[p + 1], eptr_h[p], eptr_h[p + 1]
            lei = torch.stack([src[ea:eb] - a, dst[ea:eb] - a]).contiguous()
            if eprob is not None:
                bp = eprob[ea:eb].contiguous()
            elif dev.type == "cuda" and eb > ea:
                from . import ops
                bp = ops.degree_prior(lei, b - a)
            else:
                bp = degree_prior(lei, b - a)
            self.batches.append(Batch(x=xs[a:b].contiguous(), edge_index=lei, y=ys[a:b].contiguous(), train_mask=tm[a:b].contiguous(),
                                      val_mask=vm[a:b].contiguous(), test_mask=sm[a:b].contiguous(), prob=bp, part=p,
                                      node_ids=perm[a:b]))
        self.shuffle = shuffle
        self._gen = torch.Generator().manual_seed(seed)

    def __len__(self):
        return self.num_parts

    def __getitem__(self, i):
        return self.batches[i]

    # ------------------------------------------------------------------ on-disk partition cache
    # The reference caches the METIS result through ClusterData(save_dir=...) (main.py:59-63; PyG 2.3.1 writes
    # `save_dir/partition_<P>.pt` = (adj, partptr, perm)) and re-slices every batch from it on the CPU.  Here the cache holds the
    # result of that slicing too, as ONE tensor-only safetensors file `save_dir/sgs_partitions_<P>.safetensors`:
    #   perm [N] i64        original node id of new node i (nodes grouped by partition; PyG's `perm`)
    #   node_ptr [P+1] i64  partition p owns new nodes node_ptr[p]:node_ptr[p+1]          (PyG's `partptr`)
    #   edge_ptr [P+1] i64  ... and intra-partition edges edge_ptr[p]:edge_ptr[p+1]
    #   x [N,F] f32, y [N] i64, train_mask / val_mask / test_mask [N] u8     node attributes in new order
    #   edge_index [2,E'] i64   intra-partition edges, partition-LOCAL node ids, row-sorted inside each partition
    #   prob [E'] f32           the edge prior sliced like edge_index (add_degree / add_ER output)
    # metadata: format = "sgs-partitions-v1", num_parts, dropped_edges.  Nothing in the file is executable.
    FORMAT = "sgs-partitions-v1"

    @staticmethod
    def cache_path(save_dir: str, num_parts: int) -> str:
        import os
        return os.path.join(save_dir, f"sgs_partitions_{int(num_parts)}.safetensors")

    def save(self, save_dir: str) -> str:
        import os
        from safetensors.torch import save_file
        os.makedirs(save_dir, exist_ok=True)
        cat = lambda k, dim=0: torch.cat([getattr(b, k) for b in self.batches], dim=dim).contiguous().cpu()      # noqa: E731
        t = dict(perm=self.perm.cpu(), node_ptr=self.node_ptr.cpu(), edge_ptr=self.edge_ptr.cpu(), x=cat("x"), y=cat("y"),
                 train_mask=cat("train_mask").to(torch.uint8), val_mask=cat("val_mask").to(torch.uint8),
                 test_mask=cat("test_mask").to(torch.uint8), edge_index=cat("edge_index", 1), prob=cat("prob"))
        path = self.cache_path(save_dir, self.num_parts)
        save_file(t, path, metadata={"format": self.FORMAT, "num_parts": str(self.num_parts), "dropped_edges": str(self.dropped_edges)})
        return path

    @classmethod
    def load(cls, save_dir: str, num_parts: int, device="cuda:0", shuffle=False, seed=0) -> "ResidentPartitions":
        """Rebuild the resident partition tables from `save()`'s file: memory-mapped read, one host-to-device copy per tensor,
        then per-partition views (no re-partitioning, no prior recomputation)."""
        from safetensors import safe_open
        dev = torch.device(device)
        path = cls.cache_path(save_dir, num_parts)
        with safe_open(path, framework="pt", device="cpu") as f:
            meta = f.metadata() or {}
            if meta.get("format") != cls.FORMAT or int(meta.get("num_parts", -1)) != int(num_parts):
                raise ValueError(f"{path}: not a {cls.FORMAT} cache for {num_parts} partitions")
            t = {k: f.get_tensor(k).to(dev) for k in f.keys()}
        self = cls.__new__(cls)
        P = int(num_parts)
        self.num_parts, self.perm, self.node_ptr, self.edge_ptr = P, t["perm"], t["node_ptr"], t["edge_ptr"]
        self.dropped_edges = int(meta.get("dropped_edges", 0))
        nptr, eptr = t["node_ptr"].tolist(), t["edge_ptr"].tolist()
        if len(nptr) != P + 1 or len(eptr) != P + 1 or nptr[-1] != t["x"].shape[0] or eptr[-1] != t["edge_index"].shape[1]:
            raise ValueError(f"{path}: inconsistent partition pointers")
        tm, vm, sm = t["train_mask"].bool(), t["val_mask"].bool(), t["test_mask"].bool()
        self.batches = []
        for p in range(P):
            a, b, ea, eb = nptr[p], nptr[p + 1], eptr[p], eptr[p + 1]
            self.batches.append(Batch(x=t["x"][a:b], edge_index=t["edge_index"][:, ea:eb].contiguous(), y=t["y"][a:b], train_mask=tm[a:b],
                                      val_mask=vm[a:b], test_mask=sm[a:b], prob=t["prob"][ea:eb], part=p, node_ids=t["perm"][a:b]))
        self.shuffle = shuffle
        self._gen = torch.Generator().manual_seed(seed)
        return self

    def __iter__(self):
        order = torch.randperm(self.num_parts, generator=self._gen).tolist() if self.shuffle else range(self.num_parts)
        for i in order:
            yield self.batches[i]
