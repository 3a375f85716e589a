"""GPU: grid subsampling and radius neighbours through the C ABI against the
reference's golden vectors and the CPU oracle (bit-exact bar for index work)."""
import numpy as np
import pytest
import torch

from conftest import assert_rows_equal_up_to_ties, canon_ties, load_golden
from oracle import native
from oracle.gen_golden import pairs_for
from superpoints_registration_amd import get_config, ops, synthetic
from superpoints_registration_amd.kpconv import Preprocessor

pytestmark = pytest.mark.gpu
CASES = ["ragged", "lattice", "tiny", "dense"]


@pytest.fixture(scope="module")
def gold():
    return load_golden("preprocess.npz")


def test_mfma_layout_selftest(device):
    ops.selftest()


def _cu(lens, device):
    return ops.lengths_to_cu([int(v) for v in lens], device)


@pytest.mark.parametrize("case", CASES)
def test_grid_subsample_reference_order_bit_exact(gold, device, case):
    pts, lens, dl = gold[f"{case}.pts"], gold[f"{case}.lens"], float(gold[f"{case}.dl"])
    sub, sub_lens = ops.grid_subsample(torch.from_numpy(pts).to(device), _cu(lens, device), dl,
                                       order=ops.ORDER_REFERENCE)
    assert np.array_equal(sub_lens.cpu().numpy(), gold[f"{case}.sub_lens"])
    assert np.array_equal(sub.cpu().numpy().view(np.uint32), gold[f"{case}.sub"].view(np.uint32))


@pytest.mark.parametrize("case", CASES)
def test_grid_subsample_canonical_order(gold, device, case):
    pts, lens, dl = gold[f"{case}.pts"], gold[f"{case}.lens"], float(gold[f"{case}.dl"])
    sub, sub_lens = ops.grid_subsample(torch.from_numpy(pts).to(device), _cu(lens, device), dl,
                                       order=ops.ORDER_CANONICAL)
    ref, ref_lens = native.grid_subsample(pts, lens, dl, order="canonical")
    assert np.array_equal(sub_lens.cpu().numpy(), ref_lens)
    assert np.array_equal(sub.cpu().numpy().view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("which", ["nb", "pool", "up"])
def test_radius_neighbors_vs_reference_golden(gold, device, case, which):
    pts, lens = gold[f"{case}.pts"], gold[f"{case}.lens"]
    sub, sub_lens, r = gold[f"{case}.sub"], gold[f"{case}.sub_lens"], float(gold[f"{case}.radius"])
    q, s, ql, sl, rad = {"nb": (pts, pts, lens, lens, r), "pool": (sub, pts, sub_lens, lens, r),
                         "up": (pts, sub, lens, sub_lens, 2 * r)}[which]
    ref = gold[f"{case}.{which}"].astype(np.int64)
    limit = 119
    got, mc = ops.radius_neighbors(torch.from_numpy(q).to(device), torch.from_numpy(s).to(device),
                                   _cu(ql, device), _cu(sl, device), rad, limit)
    got = got.cpu().numpy().astype(np.int64)
    assert mc == ref.shape[1]                       # untruncated row width of the reference
    w = min(mc, limit)
    assert got.shape == (q.shape[0], w)
    # the oracle is literally identical (same (d2, index) tie rule, same truncation)
    orc, _ = native.radius_neighbors(q, s, ql, sl, rad, limit=limit)
    assert np.array_equal(got, orc)
    s_ext = np.concatenate([s, np.full((1, 3), 1e6, np.float32)])
    if w == ref.shape[1]:
        assert np.array_equal(canon_ties(ref, q, s_ext)[0], canon_ties(got, q, s_ext)[0])
    assert_rows_equal_up_to_ties(ref[:, :w], got, q, s_ext, truncated=w < ref.shape[1])
    if case in ("ragged", "tiny") and which == "nb":
        assert np.array_equal(got, ref[:, :w])      # tie-free: bit-exact vs the reference itself


@pytest.mark.parametrize("case", CASES)
def test_both_search_algorithms_agree(gold, device, case):
    """algo 0 (dense cell table) and algo 1 (sorted keys + binary search) are
    the same function."""
    pts, lens, r = gold[f"{case}.pts"], gold[f"{case}.lens"], float(gold[f"{case}.radius"])
    d, cu = torch.from_numpy(pts).to(device), _cu(lens, device)
    a, ma = ops.radius_neighbors(d, d, cu, cu, r, 64, algo=0)
    b, mb = ops.radius_neighbors(d, d, cu, cu, r, 64, algo=1)
    assert ma == mb and torch.equal(a, b)


def test_cell_table_overflow_falls_back(device):
    # two far-apart clusters in one cloud: bounding box needs far more cells than the table holds
    rng = np.random.default_rng(0)
    a = rng.uniform(0, 0.05, (300, 3)).astype(np.float32)
    pts = np.concatenate([a, a + np.float32(60.0)])
    d, cu = torch.from_numpy(pts).to(device), _cu([600], device)
    got, mc = ops.radius_neighbors(d, d, cu, cu, 0.01, 32)           # silently retried with algo 1
    orc, mc2 = native.radius_neighbors(pts, pts, [600], [600], 0.01, limit=32)
    assert mc == mc2 and np.array_equal(got.cpu().numpy(), orc)


def test_radius_table_serves_several_searches_like_separate_calls(gold, device):
    """ops.RadiusTable (spr_radius_table_build / _query): ONE cell table per (supports, radius) answers the self
    search, a search from other queries and a second self search -- each row for row what spr_radius_neighbors
    (and the CPU oracle) give for the same call; the pyramid builds three tables for its seven searches."""
    pts, lens, r = gold["dense.pts"], gold["dense.lens"], float(gold["dense.radius"])
    d, cu = torch.from_numpy(pts).to(device), _cu(lens, device)
    rng = np.random.default_rng(5)
    qlens = [max(1, int(l) // 3) for l in lens]
    offs = np.concatenate([[0], np.cumsum(lens)])
    qs = np.concatenate([pts[offs[c]:offs[c + 1]][rng.permutation(int(lens[c]))[:qlens[c]]] + np.float32(1e-3)
                         for c in range(len(lens))]).astype(np.float32)
    q, qcu = torch.from_numpy(qs).to(device), _cu(qlens, device)
    table = ops.RadiusTable(d, cu, r)
    assert table.matches(d, cu, r) and not table.matches(d, cu, 2 * r) and not table.matches(q, qcu, r)
    a, ma = table.query(d, cu, 40)            # self search: cell-order walk
    b, mb = table.query(q, qcu, 40)           # other queries, same supports
    c, mc = table.query(d, cu, 25)            # another limit against the same build
    for (got, m), (qq, ql, lim) in zip(((a, ma), (b, mb), (c, mc)), ((pts, lens, 40), (qs, qlens, 40), (pts, lens, 25))):
        orc, mo = native.radius_neighbors(qq, pts, ql, lens, r, limit=lim)
        assert m == mo and np.array_equal(got.cpu().numpy(), orc)
        sep, ms = ops.radius_neighbors(torch.from_numpy(qq).to(device), d, _cu(ql, device), cu, r, lim)
        assert ms == m and torch.equal(sep, got)


def test_radius_table_overflow_and_empty_results(device):
    """A geometry the table cannot hold is answered through the sorted-key path (same rows); a query set without
    any neighbour raises like the reference (cpp_neighbors/wrapper.cpp:201-205)."""
    rng = np.random.default_rng(0)
    a = rng.uniform(0, 0.05, (300, 3)).astype(np.float32)
    pts = np.concatenate([a, a + np.float32(60.0)])
    d, cu = torch.from_numpy(pts).to(device), _cu([600], device)
    got, m = ops.RadiusTable(d, cu, 0.01).query(d, cu, 32)
    orc, mo = native.radius_neighbors(pts, pts, [600], [600], 0.01, limit=32)
    assert m == mo and np.array_equal(got.cpu().numpy(), orc)
    far = torch.full((2, 3), 100.0, device=device)
    with pytest.raises(RuntimeError):
        ops.RadiusTable(d, cu, 0.01).query(far, _cu([2], device), 8)


def test_limit_below_max_count_keeps_k_nearest(gold, device):
    pts, lens, r = gold["dense.pts"], gold["dense.lens"], float(gold["dense.radius"])
    got, mc = ops.radius_neighbors(torch.from_numpy(pts).to(device), torch.from_numpy(pts).to(device),
                                   _cu(lens, device), _cu(lens, device), r, 40)
    orc, mc2 = native.radius_neighbors(pts, pts, lens, lens, r, limit=40)
    assert mc == mc2 and np.array_equal(got.cpu().numpy(), orc)


def test_errors_follow_the_reference(device):
    # no neighbour anywhere -> RuntimeError("Error") (cpp_neighbors/wrapper.cpp:201-205)
    q = torch.zeros((2, 3), device=device)
    s = torch.full((2, 3), 100.0, device=device)
    cu = _cu([2], device)
    with pytest.raises(RuntimeError):
        ops.radius_neighbors(q, s, cu, cu, 0.1, 8)
    with pytest.raises(RuntimeError):
        ops.grid_subsample(torch.zeros((0, 3), device=device), _cu([0], device), 0.1)


def test_full_size_pair_16384(device):
    """BASELINE config 2 sizes: 2 x 16 384 points, three pyramid levels, checked
    against the CPU oracle (bit-exact) and by size-independent properties."""
    src, tgt, _ = synthetic.make_pair(16384, seed=1)
    pts = np.concatenate([src, tgt])
    lens = [16384, 16384]
    cu = _cu(lens, device)
    d_pts = torch.from_numpy(pts).to(device)
    sub, sub_lens = ops.grid_subsample(d_pts, cu, 0.05)
    ref, ref_lens = native.grid_subsample(pts, lens, 0.05)
    assert np.array_equal(sub_lens.cpu().numpy(), ref_lens)
    assert np.array_equal(sub.cpu().numpy().view(np.uint32), ref.view(np.uint32))
    nb, mc = ops.radius_neighbors(d_pts, d_pts, cu, cu, 0.0625, 40)
    nbh = nb.cpu().numpy()
    orc, mc2 = native.radius_neighbors(pts, pts, lens, lens, 0.0625, limit=40)
    assert mc == mc2 and np.array_equal(nbh, orc)
    # properties: self is the first neighbour; rows sorted by distance; same-cloud only
    assert np.array_equal(nbh[:, 0], np.arange(len(pts)))
    valid = nbh < len(pts)
    cloud = (np.arange(len(pts)) >= 16384)
    assert np.all(((nbh >= 16384) == cloud[:, None]) | ~valid)
    ext = np.concatenate([pts, np.full((1, 3), 1e6, np.float32)])
    d2 = ((pts[:, None, :] - ext[nbh]) ** 2).sum(-1)
    d2 = np.where(valid, d2, 1e30)
    assert np.all(np.diff(d2, axis=1) >= -1e-9)
    assert np.all(d2[valid] < 0.0625 ** 2 * (1 + 1e-5))


@pytest.mark.parametrize("tag", ["3dmatch", "kitti", "modelnet"])
def test_preprocessor_pyramid_matches_reference(device, tag):
    g = load_golden(f"regtr_{tag}_b2.npz")
    B = int(g["B"])
    pairs, sizes = pairs_for(tag, B)
    src = [torch.from_numpy(p[0][:n]).to(device) for p, (n, m) in zip(pairs, sizes)]
    tgt = [torch.from_numpy(p[1][:m]).to(device) for p, (n, m) in zip(pairs, sizes)]
    meta = Preprocessor(get_config(tag))(src + tgt)
    L = int(g["levels"])
    assert len(meta["points"]) == L
    for l in range(L):
        p = meta["points"][l].cpu().numpy()
        assert np.array_equal(p.view(np.uint32), g[f"points{l}"].view(np.uint32))       # order + bits
        assert np.array_equal(meta["stack_lengths"][l].cpu().numpy(), g[f"lens{l}"])
        ext = np.concatenate([p, np.full((1, 3), 1e6, np.float32)])
        for key in ("neighbors", "pools", "upsamples"):
            if f"{key}{l}" not in g:
                assert meta[key][l].shape == (0, 1)
                continue
            ref = g[f"{key}{l}"].astype(np.int64)
            got = meta[key][l].cpu().numpy()
            assert got.dtype == np.int64 and got.shape == ref.shape
            q = {"neighbors": p, "pools": g[f"points{l + 1}"] if key == "pools" else p,
                 "upsamples": p}[key]
            s = {"neighbors": p, "pools": p, "upsamples": g[f"points{l + 1}"] if key == "upsamples" else p}[key]
            s_ext = np.concatenate([s, np.full((1, 3), 1e6, np.float32)])
            # strict contract (conftest.assert_rows_equal_up_to_ties): identical d2 bits at
            # every position, identical indices outside equal-d2 runs, identical index sets
            # inside them -- for truncated matrices too (a tie may straddle the cut there)
            lim = int(get_config(tag).neighborhood_limits[l])
            n_diff, n_cut = assert_rows_equal_up_to_ties(ref, got, q, s_ext, truncated=ref.shape[1] == lim)
            # how often the tie order shows at all: conv rows tie rarely; pool / upsample queries
            # are barycentres and tie structurally (a 2-point voxel's barycentre is equidistant
            # from both points)
            assert 1.0 - n_diff / len(ref) > (0.97 if key == 'neighbors' else 0.85)
            assert n_cut <= n_diff


@pytest.mark.parametrize("case,limit", [("lattice", 8), ("lattice", 20), ("dense", 40)])
def test_truncated_rows_vs_reference_with_ties_at_the_cut(gold, device, case, limit):
    """limit far below the reference's row width: the cut lands inside equal-d2 runs on the
    lattice (structural ties).  d2 bits must still match the reference column for column."""
    pts, lens, r = gold[f"{case}.pts"], gold[f"{case}.lens"], float(gold[f"{case}.radius"])
    ref = gold[f"{case}.nb"].astype(np.int64)[:, :limit]          # kpconv.py:259-260 slice
    got, _ = ops.radius_neighbors(torch.from_numpy(pts).to(device), torch.from_numpy(pts).to(device),
                                  _cu(lens, device), _cu(lens, device), r, limit)
    s_ext = np.concatenate([pts, np.full((1, 3), 1e6, np.float32)])
    n_diff, n_cut = assert_rows_equal_up_to_ties(ref, got.cpu().numpy(), pts, s_ext, truncated=True)
    if case == "lattice":
        assert n_cut > 0, "the lattice case is meant to put ties on the cut"


def test_effect_of_a_tie_straddling_the_cut_is_one_neighbour(gold, device):
    """When an equal-d2 run straddles `limit`, the reference keeps whichever member its
    kd-tree happened to visit first; we keep the lowest index.  The two choices differ by ONE
    neighbour, so KPConv / max-pool outputs differ by at most that neighbour's contribution:
        |d out[n, :]| <= max_p w_p * |(x_a - x_b) W_p| / count   (influence w_p <= 1)
    Measured here on the lattice (every row ties) by swapping the last kept entry of every
    full row with its cut partner, and asserted against the bound."""
    pts, lens, r = gold["lattice.pts"], gold["lattice.lens"], float(gold["lattice.radius"])
    limit = 8
    d, cu = torch.from_numpy(pts).to(device), _cu(lens, device)
    wide, _ = ops.radius_neighbors(d, d, cu, cu, r, 64)
    wide = wide.cpu().numpy().astype(np.int64)
    ns = len(pts)
    s_ext = np.concatenate([pts, np.full((1, 3), 1e6, np.float32)])
    from conftest import ref_d2
    d2 = ref_d2(wide, pts, s_ext)
    ours = wide[:, :limit].copy()
    alt = ours.copy()
    straddle = (d2[:, limit - 1] == d2[:, limit]) & (wide[:, limit] != ns)
    assert straddle.sum() > 50
    alt[straddle, limit - 1] = wide[straddle, limit]                # the other member of the run
    cin, cout = 32, 32
    x = synthetic.rand((ns, cin), 5, 0.0, 1.0)
    w = synthetic.rand((15, cin, cout), 6, -0.2, 0.2)
    kp = torch.from_numpy(gold_kpts())
    ext = 0.05

    def conv(idx):
        return ops.kpconv(d, d, torch.from_numpy(idx.astype(np.int32)).to(device), x.to(device), w.to(device),
                          kp.to(device), ext, rows_sorted=True).cpu().numpy()
    ya, yb = conv(ours), conv(alt)
    delta = np.abs(ya - yb).max(1)
    assert np.all(delta[~straddle] == 0)                            # untouched rows: bit identical
    # bound: one neighbour's contribution, sum over kernel points of |(x_a - x_b) W_p| / count
    xa, xb = x.numpy()[ours[:, limit - 1]], x.numpy()[alt[:, limit - 1]]
    per_kp = np.abs(np.einsum('nc,pco->npo', xa, w.numpy())) + np.abs(np.einsum('nc,pco->npo', xb, w.numpy()))
    bound = per_kp.max(1).max(1) * 2.0 / 1.0
    assert np.all(delta <= bound + 1e-6)
    mp_a = ops.maxpool(x.to(device), torch.from_numpy(ours.astype(np.int32)).to(device)).cpu().numpy()
    mp_b = ops.maxpool(x.to(device), torch.from_numpy(alt.astype(np.int32)).to(device)).cpu().numpy()
    assert np.all(np.abs(mp_a - mp_b).max(1)[~straddle] == 0)
    assert np.all(np.abs(mp_a - mp_b) <= np.abs(xa - xb) + 1e-7)    # max over sets differing in one member


def gold_kpts():
    return load_golden("ops.npz")["kp.c32.kpts"] * np.float32(0.0625 / 0.125)


@pytest.mark.parametrize("n,voxel,extent", [(120000, 0.3, 50.0), (5000, 0.05, 1.0), (7, 10.0, 1.0)])
def test_voxel_downsample_keeps_the_first_point_of_every_voxel(device, n, voxel, extent):
    """GPU counterpart of the KITTI loader's kiss_icp voxel_down_sample
    (kitti_pred.py:12-14, :203-204): voxel = trunc(p / voxel_size) in float64 (negative
    coordinates truncate toward zero), first point per voxel kept."""
    rng = np.random.default_rng(n)
    pts = rng.uniform(-extent, extent, (n, 3)).astype(np.float32)
    pts[: n // 10] *= 0.01                                            # many points per voxel near the origin
    out = ops.voxel_downsample(torch.from_numpy(pts).to(device), voxel).cpu().numpy()
    from oracle.torch_oracle import voxel_down_sample
    ref = voxel_down_sample(pts, voxel)                               # restated kiss-icp VoxelDownsample
    assert out.shape == ref.shape and np.array_equal(out.view(np.uint32), ref.view(np.uint32))


def test_radius_search_with_more_than_65535_coincident_supports_in_range(device):
    """the overflow path of k_scan_table histograms d2 into 16-bit counters.  A raw
    dense cloud -- here 70 000 coincident returns inside one query radius, all in ONE histogram bin --
    must neither wrap the counter (now saturating) nor lose the nearest neighbours: the second pass then
    runs its replace-worst branch (far more than `cap` candidates inside the cut bin) tens of thousands
    of times per query.  Checked against a numpy brute force with the library's total order (d2, index)."""
    rng = np.random.default_rng(9)
    dup = np.tile(np.array([[0.5, 0.5, 0.5]], np.float32), (70000, 1))
    near = (np.array([[0.5, 0.5, 0.5]]) + rng.normal(0, 0.02, (300, 3))).astype(np.float32)
    supports = np.concatenate([near[:150], dup, near[150:]])            # duplicates in the middle of the index range
    queries = (np.array([[0.5, 0.5, 0.5]]) + rng.normal(0, 0.03, (64, 3))).astype(np.float32)
    radius, limit = 0.2, 40
    q, s_ = torch.from_numpy(queries).to(device), torch.from_numpy(supports).to(device)
    qcu, scu = ops.lengths_to_cu([len(queries)], device), ops.lengths_to_cu([len(supports)], device)
    out, mc = ops.radius_neighbors(q, s_, qcu, scu, radius, limit)
    out = out.cpu().numpy()
    assert mc >= 70000 and out.shape == (64, limit)
    d = queries[:, None, :] - supports[None, :, :]                        # nanoflann order: query - support
    d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
    assert d2.dtype == np.float32
    for i in range(64):
        inr = np.nonzero(d2[i] < np.float32(radius) ** 2)[0]
        order = inr[np.lexsort((inr, d2[i][inr]))][:limit]               # (d2, index) ascending, cut at limit
        assert np.array_equal(out[i], order), i


@pytest.mark.parametrize("limit", [7, 40, 64, 74, 128])
def test_wave_per_query_selection_is_the_thread_per_query_selection(device, limit):
    """Round 5: `RadiusTable.query(dense=True)` (k_knn_wave: one wave per query, the K nearest kept sorted in the wave's
    registers while it scans the coalesced record runs of the 27 cells -- no scratch rows, no sort kernel) must return
    exactly the rows of the thread-per-query path (`dense=False`: k_scan_table + rank sort), which the golden tests
    pin to the reference: self search and a search with other queries, several ragged clouds, sparse rows (a few
    supports in range), dense rows (hundreds in range against limits from 7 to 128: every merge of a full row, both
    row registers of limits > 64), equal distances (lattice points), a cloud of one point, and 3 000 coincident
    supports (ties broken by the support index through hundreds of batches)."""
    rng = np.random.default_rng(limit)
    clouds = [rng.uniform(0, 1.0, (3000, 3)),                                  # dense: ~100-300 in range
              rng.uniform(0, 6.0, (2500, 3)),                                  # sparse: 0-3 in range
              np.stack(np.meshgrid(*[np.arange(12) * 0.07] * 3, indexing="ij"), -1).reshape(-1, 3),   # lattice: ties
              np.array([[0.3, 0.3, 0.3]]),                                     # a cloud of one point
              np.concatenate([np.tile([[0.5, 0.5, 0.5]], (3000, 1)), rng.normal(0.5, 0.05, (500, 3))])]
    sup = np.concatenate(clouds).astype(np.float32)
    lens = [len(c) for c in clouds]
    qry_clouds = [c[rng.permutation(len(c))[:max(1, len(c) // 3)]] + rng.normal(0, 0.01, (max(1, len(c) // 3), 3))
                  for c in clouds]
    qry = np.concatenate(qry_clouds).astype(np.float32)
    s_, q = torch.from_numpy(sup).to(device), torch.from_numpy(qry).to(device)
    scu, qcu = ops.lengths_to_cu(lens, device), ops.lengths_to_cu([len(c) for c in qry_clouds], device)
    radius = 0.2
    for queries, cu in ((s_, scu), (q, qcu)):
        t0 = ops.RadiusTable(s_, scu, radius)
        a, ma = t0.query(queries, cu, limit, dense=False)
        b, mb = t0.query(queries, cu, limit, dense=True)
        assert ma == mb and a.shape == b.shape
        assert torch.equal(a, b)
    # ... and against a numpy brute force with the library's total order (d2, index) on the dense cloud
    d = qry[:200, None, :] - sup[None, :lens[0], :]
    d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
    got = b[:200].cpu().numpy()
    for i in range(200):
        inr = np.nonzero(d2[i] < np.float32(radius) ** 2)[0]
        order = inr[np.lexsort((inr, d2[i][inr]))][:limit]
        assert np.array_equal(got[i, :len(order)], order), i
        assert (got[i, len(order):] == len(sup)).all()


@pytest.mark.gpu
def test_cell_order_is_a_spatial_permutation_and_the_ordered_maxpool_is_the_maxpool(device):
    """ops.cell_order: a permutation of the rows, clouds kept in batch order, consecutive rows close in space;
    ops.maxpool(order=...) walks its queries in that order and must return the same bits (round 5: the walk
    order only decides which rows share L2)."""
    lens = [3000, 1, 777, 5000]
    g = torch.Generator().manual_seed(3)
    pts = torch.cat([torch.rand((n, 3), generator=g) * 2.0 + 5.0 * i for i, n in enumerate(lens)]).to(device)
    cu = ops.lengths_to_cu(lens, device)
    order = ops.cell_order(pts, cu, 0.1)
    o = order.cpu().numpy()
    assert o.dtype == np.int32 and sorted(o.tolist()) == list(range(sum(lens)))
    offs = np.concatenate([[0], np.cumsum(lens)])
    for i in range(len(lens)):
        seg = o[offs[i]:offs[i + 1]]
        assert seg.min() >= offs[i] and seg.max() < offs[i + 1]          # clouds stay in batch order
    p = pts.cpu().numpy()
    walk = np.linalg.norm(np.diff(p[o[:3000]], axis=0), axis=1).mean()
    stored = np.linalg.norm(np.diff(p[:3000], axis=0), axis=1).mean()
    assert walk < 0.3 * stored, (walk, stored)                            # neighbours in the walk are neighbours in space
    nq, ns, c, k = sum(lens), 20000, 64, 11
    x = torch.randn((ns, c), generator=g).to(device)
    idx = torch.randint(0, ns + 1, (nq, k), generator=g, dtype=torch.int32).to(device)   # ns = the shadow row
    a = ops.maxpool(x, idx)
    b = ops.maxpool(x, idx, order)
    assert torch.equal(a, b)
