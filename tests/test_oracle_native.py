"""CPU: the C restatement (oracle/spr_oracle.c) against the golden vectors the
reference's own C++ produced, and against oracle/_ref live when it is built."""
import numpy as np
import pytest

from conftest import canon_ties, load_golden
from oracle import native

CASES = ["ragged", "lattice", "tiny", "dense"]


@pytest.fixture(scope="module")
def gold():
    return load_golden("preprocess.npz")


@pytest.mark.parametrize("case", CASES)
def test_grid_subsample_bit_exact_including_order(gold, case):
    pts, lens, dl = gold[f"{case}.pts"], gold[f"{case}.lens"], float(gold[f"{case}.dl"])
    sub, sub_lens = native.grid_subsample(pts, lens, dl, order="reference")
    assert np.array_equal(sub_lens, gold[f"{case}.sub_lens"])
    assert np.array_equal(sub.view(np.uint32), gold[f"{case}.sub"].view(np.uint32))


@pytest.mark.parametrize("case", CASES)
def test_grid_subsample_canonical_is_a_permutation(gold, case):
    pts, lens, dl = gold[f"{case}.pts"], gold[f"{case}.lens"], float(gold[f"{case}.dl"])
    sub, sub_lens, keys, _ = native.grid_subsample(pts, lens, dl, order="canonical", return_keys=True)
    assert np.array_equal(sub_lens, gold[f"{case}.sub_lens"])
    off = 0
    for n in sub_lens:
        k = keys[off:off + n]
        assert np.all(k[1:] > k[:-1])                      # ascending voxel key per cloud
        a = {r.tobytes() for r in sub[off:off + n]}
        b = {r.tobytes() for r in gold[f"{case}.sub"][off:off + n]}
        assert a == b                                      # same barycentres, bit for bit
        off += n


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("which", ["nb", "pool", "up"])
def test_radius_neighbors_vs_reference(gold, case, which):
    pts, lens = gold[f"{case}.pts"], gold[f"{case}.lens"]
    sub, sub_lens, r = gold[f"{case}.sub"], gold[f"{case}.sub_lens"], float(gold[f"{case}.radius"])
    q, s, ql, sl, rad = {"nb": (pts, pts, lens, lens, r), "pool": (sub, pts, sub_lens, lens, r),
                         "up": (pts, sub, lens, sub_lens, 2 * r)}[which]
    ref = gold[f"{case}.{which}"].astype(np.int64)
    got, max_count = native.radius_neighbors(q, s, ql, sl, rad, limit=0)
    assert got.shape == ref.shape and max_count == ref.shape[1]
    s_ext = np.concatenate([s, np.full((1, 3), 1e6, np.float32)])
    ci, cd = canon_ties(ref, q, s_ext)
    gi, gd = canon_ties(got, q, s_ext)
    assert np.array_equal(cd.view(np.uint32), gd.view(np.uint32))   # same d2 bits per column
    assert np.array_equal(ci, gi)                                     # same indices after tie canonicalisation
    exact_rows = (ref == got).all(1).mean()
    if case in ("ragged", "tiny") and which == "nb":
        assert exact_rows == 1.0                                      # tie-free data: literally identical
        # (pool / upsample queries are barycentres: a 2-point voxel's barycentre is
        # exactly equidistant from both points, so those rows tie structurally)
    elif case == "dense" and which == "nb":
        assert exact_rows > 0.99                                      # a few exact float32 d2 ties


def test_limit_truncation_keeps_nearest(gold):
    pts, lens, r = gold["dense.pts"], gold["dense.lens"], float(gold["dense.radius"])
    full, mc = native.radius_neighbors(pts, pts, lens, lens, r, limit=0)
    cut, mc2 = native.radius_neighbors(pts, pts, lens, lens, r, limit=40)
    assert mc == mc2 and cut.shape[1] == 40 and mc > 40
    assert np.array_equal(cut, full[:, :40])


def test_umap_order_matches_std_unordered_map_model():
    # epoch/sort model used by the HIP kernel == sequential libstdc++ model in the oracle
    sched = [1, 13, 29, 59, 127, 257, 541, 1109, 2357, 5087, 10273, 20753]
    rng = np.random.default_rng(0)
    for n in [1, 13, 14, 30, 257, 258, 3000, 9483]:
        keys = rng.choice(1 << 22, n, replace=False).astype(np.uint64)
        lst, done, e = np.zeros(0, np.int64), 0, 0
        while done < n:
            e += 1
            nb = sched[e]
            take = min(n, nb) - done
            arr = np.concatenate([lst, np.arange(done, done + take)])
            bk = (keys[arr] % np.uint64(nb)).astype(np.int64)
            pos = np.arange(len(arr))
            first = np.full(nb, 1 << 60, np.int64)
            np.minimum.at(first, bk, pos)
            lst = arr[np.lexsort((-pos, -first[bk]))]
            done += take
        assert np.array_equal(native.umap_order(keys), lst)


@pytest.mark.skipif(not native.ref_available(), reason="oracle/_ref not built (no /root/reference)")
def test_oracle_vs_compiled_reference_live():
    rng = np.random.default_rng(5)
    for n, dl in [(3000, 0.05), (3000, 0.1), (800, 0.025)]:
        pts = rng.uniform(0, 1.5, (2 * n + 11, 3)).astype(np.float32)
        pts[:, 2] *= 0.05
        lens = [n, n + 11]
        a, al = native.ref_grid_subsample(pts, lens, dl)
        b, bl = native.grid_subsample(pts, lens, dl)
        assert np.array_equal(al, bl) and np.array_equal(a.view(np.uint32), b.view(np.uint32))
        ra = native.ref_radius_neighbors(pts, pts, lens, lens, 2.5 * dl)
        rb, _ = native.radius_neighbors(pts, pts, lens, lens, 2.5 * dl)
        s_ext = np.concatenate([pts, np.full((1, 3), 1e6, np.float32)])
        assert np.array_equal(canon_ties(ra, pts, s_ext)[0], canon_ties(rb, pts, s_ext)[0])


@pytest.mark.parametrize("case", ["ragged", "lattice", "tiny", "dense"])
@pytest.mark.parametrize("limit", [8, 20, 40, 119])
def test_oracle_rows_equal_reference_up_to_ties_incl_truncation(case, limit):
    """Strict neighbour contract (conftest.assert_rows_equal_up_to_ties) of the C restatement
    against the reference's own output, also when `limit` cuts through equal-d2 runs."""
    import numpy as np
    from conftest import assert_rows_equal_up_to_ties, load_golden
    from oracle import native
    g = load_golden("preprocess.npz")
    pts, lens, r = g[f"{case}.pts"], g[f"{case}.lens"], float(g[f"{case}.radius"])
    ref = g[f"{case}.nb"].astype(np.int64)
    s_ext = np.concatenate([pts, np.full((1, 3), 1e6, np.float32)])
    orc, _ = native.radius_neighbors(pts, pts, lens, lens, r, limit=limit)
    w = min(limit, ref.shape[1])
    assert_rows_equal_up_to_ties(ref[:, :w], orc, pts, s_ext, truncated=w < ref.shape[1])
