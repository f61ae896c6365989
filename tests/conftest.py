import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "end-to-end-self-supervised-slam_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU oracle's fp32 rounding depends on torch's thread count (its convolutions and reductions split their sums by thread): with 16
    # threads instead of the GPU box's default the fp32-vs-fp32 gradient comparisons of tests/test_gpu_driver.py move from <= 1e-4 to 2.4e-3
    # on single tensors (longer sequential partial sums on the ORACLE's side).  No test may therefore change the count for the tests after it
    # -- tests/test_gpu_network.py did until round 4 (it lowered it to 16 for speed and never restored it), which made the driver tests pass
    # in alphabetical order and fail when they ran after it.  E2E_ORACLE_THREADS=n pins a count for the whole session (diagnostics).
    want = int(os.environ.get("E2E_ORACLE_THREADS", "0"))
    if want > 0:
        torch.set_num_threads(min(want, torch.get_num_threads()))


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    out = {}
    for k in z.files:
        a = z[k]
        out[k] = torch.from_numpy(a) if a.dtype.kind in "fiub" else a
    return out


@pytest.fixture
def golden():
    return load_golden
