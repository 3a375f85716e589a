import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


@pytest.fixture(scope="session")
def device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def canon_ties(idx_rows, q, s_ext):
    """Canonicalise neighbour rows for comparison: the reference orders equal-d2
    runs by kd-tree visit order (std::sort is not stable), the oracle and the
    HIP kernel by index.  Sort every row by (d2, index) with the reference's
    float32 d2 arithmetic; shadow entries (index == Ns) sort last."""
    idx_rows = np.asarray(idx_rows, np.int64)
    ns = s_ext.shape[0] - 1
    d = q[:, None, :].astype(np.float32) - s_ext[idx_rows].astype(np.float32)
    d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]).astype(np.float32)
    d2 = (d2 + d[..., 2] * d[..., 2]).astype(np.float32)
    d2 = np.where(idx_rows == ns, np.float32(np.inf), d2)
    order = np.lexsort((idx_rows, d2), axis=1)
    return np.take_along_axis(idx_rows, order, 1), np.take_along_axis(d2, order, 1)
