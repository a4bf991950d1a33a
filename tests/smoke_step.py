"""One tiny invocation of the hot path on cuda:0 checked against the oracle (used by
__graft_entry__.smoke)."""
import torch
import torch.nn.functional as F

from oracle import sgs_oracle as O


def run_smoke(pkg):
    ops = pkg.ops
    dev = "cuda:0"
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
