"""Inputs of the production-size golden fixtures, generated from INTEGER hashes only (numpy uint64 arithmetic, one
float64 division per value): the same bits in the build container (where tests/golden/gen_golden.py runs the reference on
them) and on the GPU box (where the parity tests rebuild them), whatever the CPU's vector ISA.  Nothing here comes from
the reference; the fixtures store only what the reference computed from these inputs."""
import numpy as np
import torch

_M32 = np.uint64(0xFFFFFFFF)


def hash32(idx, salt: int):
    """lowbias32-style avalanche of (idx, salt) -> uint64 array with values in [0, 2^32)."""
    x = (np.asarray(idx, dtype=np.uint64) + np.uint64((salt * 0x9E3779B1) & 0xFFFFFFFF)) & _M32
    x = ((x ^ (x >> np.uint64(16))) * np.uint64(0x7FEB352D)) & _M32
    x = ((x ^ (x >> np.uint64(15))) * np.uint64(0x846CA68B)) & _M32
    return x ^ (x >> np.uint64(16))


def uniform_grid(n: int, salt: int, levels: int = 4001, lo: float = -1.0, hi: float = 1.0) -> torch.Tensor:
    """n float32 values on a uniform grid of `levels` points in [lo, hi]."""
    h = hash32(np.arange(n, dtype=np.uint64), salt) % np.uint64(levels)
    v = lo + (hi - lo) * (h.astype(np.float64) / float(levels - 1))
    return torch.from_numpy(v.astype(np.float32))


def skewed_graph(n: int, n_directed_target: int, salt: int) -> torch.Tensor:
    """Undirected, loop-free, coalesced, row-sorted edge_index [2,E] int64 with a heavy-tailed degree profile (endpoint =
    floor(n * u1 * u2) for two hashed uniforms), both directions stored; E is close to `n_directed_target`."""
    m = n_directed_target // 2
    keys = np.zeros(0, dtype=np.uint64)
    rnd = 0
    while keys.size < m:
        k = int((m - keys.size) * 1.5) + 64
        i = np.arange(k, dtype=np.uint64) + np.uint64(rnd * 0x1000000)
        u = [hash32(i, salt + 7 * j + 1) for j in range(4)]
        a = (((u[0] * u[1]) >> np.uint64(32)) * np.uint64(n)) >> np.uint64(32)
        b = (((u[2] * u[3]) >> np.uint64(32)) * np.uint64(n)) >> np.uint64(32)
        ok = a != b
        lo, hi = np.minimum(a[ok], b[ok]), np.maximum(a[ok], b[ok])
        keys = np.unique(np.concatenate([keys, lo * np.uint64(n) + hi]))
        rnd += 1
    # thin to m pairs by hash order (deterministic)
    order = np.argsort(hash32(keys, salt + 99), kind="stable")
    keys = keys[order[:m]]
    lo, hi = keys // np.uint64(n), keys % np.uint64(n)
    both = np.unique(np.concatenate([lo * np.uint64(n) + hi, hi * np.uint64(n) + lo])).astype(np.int64)
    return torch.from_numpy(np.stack([both // n, both % n]))


def make_partition(n: int, nfeat: int, ncls: int, n_edges_target: int, salt: int, train_frac: float = 0.66):
    """dict(x, edge_index, y, train_mask) of one synthetic partition (Reddit-like shapes when n=1013, nfeat=602, ncls=41)."""
    ei = skewed_graph(n, n_edges_target, salt)
    y = torch.from_numpy((hash32(np.arange(n, dtype=np.uint64), salt + 1000) % np.uint64(ncls)).astype(np.int64))
    x = uniform_grid(n * nfeat, salt + 2000, lo=-2.0, hi=2.0).reshape(n, nfeat).clone()
    x[torch.arange(n), y % nfeat] += 1.5
    r = hash32(np.arange(n, dtype=np.uint64), salt + 3000) % np.uint64(10000)
    tm = torch.from_numpy(r < np.uint64(int(train_frac * 10000)))
    return dict(x=x, edge_index=ei, y=y, train_mask=tm)


def degree_prior(edge_index: torch.Tensor, n: int) -> torch.Tensor:
    """The formula of the reference's add_degree (datasets.py:141-156) evaluated in float64 and rounded once to float32, so that
    the last-ulp differences between vectorised float32 exp implementations cannot reach the fixture's input."""
    row, col = edge_index[0], edge_index[1]
    E = edge_index.shape[1]
    rowcount = torch.bincount(row, minlength=n).double()
    colcount = torch.bincount(col, minlength=n).double()
    logit = (1.0 / ((colcount[row] + rowcount[col]) + 1e-10)) * E ** -0.5
    return torch.softmax(logit, dim=0).float()


def init_state(shapes: dict, salt: int) -> dict:
    """Portable initial state_dict: weight matrices glorot-like uniform(+-sqrt(6/(fan_in+fan_out))), vectors uniform(+-0.05)."""
    out = {}
    for j, (k, shp) in enumerate(sorted(shapes.items())):
        numel = int(np.prod(shp))
        bound = (6.0 / (shp[0] + shp[1])) ** 0.5 if len(shp) == 2 else 0.05
        out[k] = uniform_grid(numel, salt + 17 * j, levels=20001, lo=-bound, hi=bound).reshape(shp).clone()
    return out


def pack_mask(mask: torch.Tensor) -> torch.Tensor:
    return torch.from_numpy(np.packbits(mask.numpy().astype(np.uint8)))


def unpack_mask(packed: torch.Tensor, n: int) -> torch.Tensor:
    return torch.from_numpy(np.unpackbits(packed.numpy())[:n].astype(bool))
