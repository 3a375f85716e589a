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


def ref_d2(idx_rows, q, s_ext):
    """float32 d2 of every (query, neighbour) entry with the reference's arithmetic
    ((dx*dx + dy*dy) + dz*dz, nanoflann.hpp:432-440); shadow entries -> +inf."""
    idx_rows = np.asarray(idx_rows, np.int64)
    ns = s_ext.shape[0] - 1
    d = q[:, None, :].astype(np.float32) - s_ext[idx_rows].astype(np.float32)
    d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]).astype(np.float32)
    d2 = (d2 + d[..., 2] * d[..., 2]).astype(np.float32)
    return np.where(idx_rows == ns, np.float32(np.inf), d2)


def assert_rows_equal_up_to_ties(ref, got, q, s_ext, truncated):
    """The strict neighbour-matrix contract against the REFERENCE (nanoflann orders equal-d2
    entries by kd-tree visit order through an unstable std::sort, nanoflann.hpp:1283-1288;
    we order them by index):
      1. every row holds the same d2 values, bit for bit and position by position (both are
         sorted ascending) -- also when the matrix was cut to `limit` columns: the K-nearest
         cut keeps the K smallest d2 whatever the order inside a tie;
      2. the indices agree at every position whose d2 is unique in its row -- except, in a
         truncated matrix, a full row's entries that share the row's LAST d2 (their tie
         partner may sit just behind the cut);
      3. inside an equal-d2 run the two rows hold the same index SET (again up to the cut).
    Returns (#rows with any positional difference, #rows whose difference touches the cut)."""
    ref, got = np.asarray(ref, np.int64), np.asarray(got, np.int64)
    assert ref.shape == got.shape
    ns = s_ext.shape[0] - 1
    dr, dg = ref_d2(ref, q, s_ext), ref_d2(got, q, s_ext)
    assert np.array_equal(dr.view(np.uint32), dg.view(np.uint32)), "d2 multisets differ"
    assert np.all(dg[:, 1:] >= dg[:, :-1])                        # ascending (shadow = +inf last)
    w = ref.shape[1]
    prev_eq = np.zeros_like(dr, bool)
    prev_eq[:, 1:] = dr[:, 1:] == dr[:, :-1]
    next_eq = np.zeros_like(dr, bool)
    next_eq[:, :-1] = prev_eq[:, 1:]
    at_cut = np.zeros_like(dr, bool)
    if truncated:
        full = ref[:, -1] != ns
        at_cut = (dr == dr[:, -1:]) & full[:, None]
    unique = ~(prev_eq | next_eq) & ~at_cut
    assert np.array_equal(ref[unique], got[unique]), "indices differ outside equal-d2 runs"
    # equal-d2 runs: same index sets (compare canonically sorted rows away from the cut)
    order_r = np.lexsort((ref, dr), axis=1)
    order_g = np.lexsort((got, dg), axis=1)
    cr, cg = np.take_along_axis(ref, order_r, 1), np.take_along_axis(got, order_g, 1)
    cut_sorted = np.take_along_axis(at_cut, order_r, 1)
    assert np.array_equal(cr[~cut_sorted], cg[~cut_sorted]), "tie runs hold different index sets"
    diff_rows = (ref != got).any(1)
    cut_rows = ((ref != got) & at_cut).any(1)
    assert w >= 1
    return int(diff_rows.sum()), int(cut_rows.sum())
