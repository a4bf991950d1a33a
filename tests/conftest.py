import os
import sys

import pytest

# every test runs with the scratch-arena stream guard on (ops.workspace: a slot asked for from a stream that does not own it raises)
os.environ.setdefault("SGS_WS_GUARD", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests are skipped (not failed) when no device is visible, e.g. in the build container.
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def load_golden(name):
    import torch
    return torch.load(os.path.join(GOLDEN, name), weights_only=True)


def load_golden_fullsize(name):
    """A production-size fixture (tests/golden/gen_golden.py: gen_pipeline_fullsize): the stored part is what the REFERENCE computed;
    the inputs and the initial state are rebuilt here from integer hashes (tests/golden/portable.py) and checked against the
    stored checksums.  Returns the fixture in the small fixtures' layout (x, edge_index, y, train_mask, prob, state0, steps)."""
    import torch
    sys.path.insert(0, GOLDEN)
    import portable as PT
    fx = load_golden(name)
    part = PT.make_partition(fx["n"], fx["nfeat"], fx["ncls"], fx["e_target"], fx["salt"])
    ei = part["edge_index"]
    assert ei.shape[1] == fx["E"] and int(ei.sum()) == fx["ei_checksum"], "portable graph generator gave different bits on this host"
    assert float(part["x"].double().sum()) == fx["x_checksum"], "portable feature generator gave different bits on this host"
    F_, H, C = fx["nfeat"], fx["hid"], fx["ncls"]
    shapes = {"edge_prob_mlp.gcn1.bias": (H,), "edge_prob_mlp.gcn1.lin.weight": (H, F_), "edge_prob_mlp.gcn2.bias": (H,),
              "edge_prob_mlp.gcn2.lin.weight": (H, H), "edge_prob_mlp.fc1.weight": (H, 2 * H), "edge_prob_mlp.fc1.bias": (H,),
              "edge_prob_mlp.fc2.weight": (1, H), "edge_prob_mlp.fc2.bias": (1,), "gcn1.bias": (H,), "gcn1.lin.weight": (H, F_),
              "gcn2.bias": (C,), "gcn2.lin.weight": (C, H)}
    sd0 = PT.init_state(shapes, fx["salt"] + 5000)
    assert float(sum(v.double().abs().sum() for v in sd0.values())) == fx["state0_checksum"]
    prob = PT.degree_prior(ei, fx["n"])
    assert float(prob.double().sum()) == fx["prob_checksum"], "portable prior gave different bits on this host"
    fx.update(x=part["x"], edge_index=ei, y=part["y"], train_mask=part["train_mask"], prob=prob, state0=sd0)
    for st in fx["steps"]:
        st["mask"] = PT.unpack_mask(st["mask_packed"], fx["E"])
        st["prior_mask"] = PT.unpack_mask(st["prior_mask_packed"], fx["E"])
    return fx


# ---------------------------------------------------------------------------------------------------
# SGS_POISON=1: every buffer the host layer hands to the kernels uninitialised (torch.empty / empty_like in
# sgs_gnn_amd.ops and the scratch arena) is first filled with 0xFF bytes (NaN floats, -1 indices), so a kernel
# that reads memory it was supposed to write first shows up as a parity failure instead of passing by luck.
class _PoisonTorch:
    def __init__(self, real):
        self._real = real

    def __getattr__(self, name):
        return getattr(self._real, name)

    def empty(self, *a, **k):
        t = self._real.empty(*a, **k)
        if t.is_cuda and t.numel():
            t.view(self._real.uint8).fill_(0xFF)
        return t

    def empty_like(self, x, **k):
        t = self._real.empty_like(x, **k)
        if t.is_cuda and t.numel():
            t.view(self._real.uint8).fill_(0xFF)
        return t


@pytest.fixture(scope="session", autouse=True)
def _sgs_poison():
    if os.environ.get("SGS_POISON") != "1":
        yield
        return
    import torch
    if not torch.cuda.is_available():
        yield
        return
    import sgs_gnn_amd
    ops = sgs_gnn_amd.ops
    real_ws = ops.workspace

    def poisoned_ws(nbytes, device):
        ws = real_ws(nbytes, device)
        ws.fill_(0xFF)
        return ws

    ops.torch = _PoisonTorch(torch)
    ops.workspace = poisoned_ws
    yield
    ops.torch = torch
    ops.workspace = real_ws
