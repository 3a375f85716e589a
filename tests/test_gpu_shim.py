"""GPU: the native-level drop-in of INTEGRATION.md section 2 (superpoints_registration_amd/hip_shim.py)
called exactly the way the reference's kpconv.py calls its CPython extensions -- numpy in, numpy out,
keyword names, RuntimeError on bad input -- and compared with the reference's own C++ (oracle/_ref)."""
import numpy as np
import pytest

from conftest import assert_rows_equal_up_to_ties
from oracle import native
from superpoints_registration_amd import synthetic
from superpoints_registration_amd.hip_shim import cpp_neighbors, cpp_subsampling

pytestmark = pytest.mark.gpu


def _clouds():
    rng = np.random.default_rng(5)
    a = (synthetic.box_faces(3000, rng, 1.0) + rng.normal(0, 0.004, (3000, 3))).astype(np.float32)
    b = (synthetic.box_faces(2400, rng, 0.8) + rng.normal(0, 0.004, (2400, 3))).astype(np.float32)
    return np.concatenate([a, b]), np.array([3000, 2400], np.int32)


def test_subsample_batch_like_the_reference_extension(device):
    pts, lens = _clouds()
    sub, sub_lens = cpp_subsampling.subsample_batch(pts, lens, sampleDl=0.05, max_p=0, verbose=0)   # kpconv.py:179-183
    assert isinstance(sub, np.ndarray) and sub.dtype == np.float32 and sub_lens.dtype == np.int32
    ref, ref_lens = native.ref_grid_subsample(pts, lens, 0.05, 0)
    assert np.array_equal(sub_lens, ref_lens)
    assert np.array_equal(sub.view(np.uint32), ref.view(np.uint32))          # bit-exact barycentres AND order
    with pytest.raises(RuntimeError):
        cpp_subsampling.subsample_batch(np.zeros((0, 3), np.float32), np.array([0], np.int32), sampleDl=0.05)


def test_batch_query_like_the_reference_extension(device):
    pts, lens = _clouds()
    sub, sub_lens = cpp_subsampling.subsample_batch(pts, lens, sampleDl=0.05)
    nb = cpp_neighbors.batch_query(sub, pts, sub_lens, lens, radius=0.09)     # kpconv.py:258 (pools: queries != supports)
    assert isinstance(nb, np.ndarray) and nb.dtype == np.int32 and nb.shape[0] == sub.shape[0]
    ref = native.ref_radius_neighbors(sub, pts, sub_lens, lens, 0.09)
    assert nb.shape == ref.shape
    s_ext = np.concatenate([pts, np.full((1, 3), 1e6, np.float32)])
    assert_rows_equal_up_to_ties(ref, nb, sub, s_ext, truncated=False)
    with pytest.raises(RuntimeError):
        cpp_neighbors.batch_query(np.zeros((0, 3), np.float32), pts, np.array([0, 0], np.int32), lens, radius=0.09)


def test_shim_inside_a_non_default_stream_context(device):
    """the kernels run on the caller's current stream (uploads, kernels and read-back ordered on it)."""
    import torch
    pts, lens = _clouds()
    ref, ref_lens = native.ref_grid_subsample(pts, lens, 0.05, 0)
    side = torch.cuda.Stream(device)
    with torch.cuda.stream(side):
        for _ in range(3):                                                   # repeated: a race would show as a mismatch
            sub, sub_lens = cpp_subsampling.subsample_batch(pts, lens, sampleDl=0.05)
            assert np.array_equal(sub_lens, ref_lens) and np.array_equal(sub.view(np.uint32), ref.view(np.uint32))
            nb = cpp_neighbors.batch_query(sub, pts, sub_lens, lens, radius=0.09)
            assert nb.shape == native.ref_radius_neighbors(sub, pts, sub_lens, lens, 0.09).shape
