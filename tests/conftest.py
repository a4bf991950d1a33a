import os
import sys

import pytest

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
