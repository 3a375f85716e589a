"""GPU: the other BASELINE.json configurations as parity cases (configs[2..4]).

  configs[2]  3DMatch-shaped indoor pairs (~20 k pts, 0.025 m voxel), batch 8
  configs[3]  KITTI-shaped outdoor pairs (~120 k pts, 0.3 m voxel)
  configs[4]  ModelNet-shaped partial-overlap pairs (1 024 pts), batch 256

At these sizes the full CPU oracle is too slow for a test, so index work is
checked bit-exactly on SAMPLED query rows against a brute-force float32
restatement of the reference rule (nanoflann.hpp:249,432-440: d2 = dx*dx +
dy*dy + dz*dz in float32, strict d2 < r*r, same cloud only; rows ordered by
(d2, index) -- the tie order documented in DESIGN.md section 3), subsampling
against the C oracle, and the model output through size-independent
properties (proper rotations, determinism)."""
import numpy as np
import pytest
import torch

from oracle import native
from superpoints_registration_amd import get_config, ops, synthetic
from superpoints_registration_amd.kpconv import Preprocessor
from superpoints_registration_amd.regtr import RegTR

pytestmark = pytest.mark.gpu


def _brute_rows(q, s, q_cloud, s_cloud, rows, radius, limit):
    """Expected neighbour rows (global support indices, padded with len(s)) of the sampled queries."""
    ns = len(s)
    r2 = np.float32(radius) * np.float32(radius)
    out = np.full((len(rows), limit), ns, np.int64)
    counts = np.zeros(len(rows), np.int64)
    for o, i in enumerate(rows):
        cand = np.nonzero(s_cloud == q_cloud[i])[0]
        d = q[i][None, :].astype(np.float32) - s[cand].astype(np.float32)
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(np.float32)
        d2 = (d2 + d[:, 2] * d[:, 2]).astype(np.float32)
        keep = d2 < r2
        cand, d2 = cand[keep], d2[keep]
        order = np.lexsort((cand, d2))
        counts[o] = len(cand)
        k = min(limit, len(cand))
        out[o, :k] = cand[order[:k]]
    return out, counts


def _cloud_ids(lens):
    return np.repeat(np.arange(len(lens)), lens)


def _check_pyramid(pts_list, cfg, device, n_sample, seed):
    """Runs the Preprocessor and checks every level's matrices on sampled rows."""
    rng = np.random.default_rng(seed)
    meta = Preprocessor(cfg)([torch.from_numpy(p).to(device) for p in pts_list])
    r = cfg.first_subsampling_dl * cfg.conv_radius
    nlev = len(meta['points'])
    for l in range(nlev):
        pts = meta['points'][l].cpu().numpy()
        lens = meta['stack_lengths'][l].cpu().numpy()
        cid = _cloud_ids(lens)
        limit = int(cfg.neighborhood_limits[l])
        nb = meta['neighbors'][l].cpu().numpy()
        rows = rng.choice(len(pts), size=min(n_sample, len(pts)), replace=False)
        exp, cnt = _brute_rows(pts, pts, cid, cid, rows, r, limit)
        w = nb.shape[1]
        assert w <= limit
        assert np.array_equal(nb[rows], exp[:, :w]), f"level {l} conv neighbours"
        assert np.all(exp[:, w:] == len(pts)), f"level {l}: reference width would be larger"
        if l + 1 < nlev:
            sub = meta['points'][l + 1].cpu().numpy()
            sub_lens = meta['stack_lengths'][l + 1].cpu().numpy()
            # subsampling: bit-exact barycentres in the reference's order (C oracle)
            ref, ref_lens = native.grid_subsample(pts, lens, 2 * r / cfg.conv_radius)
            assert np.array_equal(sub_lens, ref_lens)
            assert np.array_equal(sub.view(np.uint32), ref.view(np.uint32))
            scid = _cloud_ids(sub_lens)
            pool = meta['pools'][l].cpu().numpy()
            rows = rng.choice(len(sub), size=min(n_sample, len(sub)), replace=False)
            exp, _ = _brute_rows(sub, pts, scid, cid, rows, r, limit)
            assert np.array_equal(pool[rows], exp[:, :pool.shape[1]]), f"level {l} pools"
            up = meta['upsamples'][l].cpu().numpy()
            rows = rng.choice(len(pts), size=min(n_sample, len(pts)), replace=False)
            exp, _ = _brute_rows(pts, sub, cid, scid, rows, 2 * r, limit)
            assert np.array_equal(up[rows], exp[:, :up.shape[1]]), f"level {l} upsamples"
        r *= 2
    return meta


def _check_poses(out, npairs):
    pose = out['pose'].float().cpu().numpy().reshape(-1, 3, 4)[-npairs:]
    assert np.all(np.isfinite(pose))
    R = pose[:, :, :3].astype(np.float64)
    eye = np.einsum('bij,bkj->bik', R, R)
    assert np.abs(eye - np.eye(3)).max() < 1e-4          # orthonormal
    assert np.abs(np.linalg.det(R) - 1.0).max() < 1e-4   # proper rotation (se3_torch.py:150-157)


def _model(tag, device):
    cfg = get_config(tag)
    model = RegTR(cfg)
    synthetic.fill_parameters(model, 0)
    return cfg, model.to(device).eval()


def test_config2_3dmatch_shaped_batch8(device):
    cfg, model = _model('3dmatch', device)
    pairs = [synthetic.make_pair(20000 + 137 * i, seed=10 + i) for i in range(8)]   # ragged sizes
    pts = [p[0] for p in pairs] + [p[1] for p in pairs]
    _check_pyramid(pts, cfg, device, n_sample=96, seed=2)
    batch = {"src_xyz": [torch.from_numpy(p[0]).to(device) for p in pairs],
             "tgt_xyz": [torch.from_numpy(p[1]).to(device) for p in pairs]}
    with torch.no_grad():
        o1 = model(batch)
        o2 = model(batch)
    _check_poses(o1, 8)
    assert torch.equal(o1['pose'], o2['pose'])


def test_config3_kitti_shaped_120k(device):
    cfg, model = _model('kitti', device)
    # outdoor scale: the box is stretched so that the 0.3 m voxel grid keeps ~40 k points at level 1
    pair = synthetic.make_pair(120000, seed=21, extent=60.0, jitter=0.03, trans=(1.5, -0.7, 0.1))
    _check_pyramid([pair[0], pair[1]], cfg, device, n_sample=64, seed=3)
    batch = {"src_xyz": [torch.from_numpy(pair[0]).to(device)], "tgt_xyz": [torch.from_numpy(pair[1]).to(device)]}
    with torch.no_grad():
        o1 = model(batch)
        o2 = model(batch)
    _check_poses(o1, 1)
    assert torch.equal(o1['pose'], o2['pose'])


def test_config4_modelnet_shaped_batch256(device):
    cfg, model = _model('modelnet', device)
    pairs = [synthetic.make_sphere_pair(1024, seed=100 + i) for i in range(256)]
    pts = [p[0] for p in pairs] + [p[1] for p in pairs]
    _check_pyramid(pts, cfg, device, n_sample=128, seed=4)
    batch = {"src_xyz": [torch.from_numpy(p[0]).to(device) for p in pairs],
             "tgt_xyz": [torch.from_numpy(p[1]).to(device) for p in pairs]}
    with torch.no_grad():
        o1 = model(batch)
        o2 = model(batch)
    _check_poses(o1, 256)
    assert torch.equal(o1['pose'], o2['pose'])


# ---- float parity at (close to) full size: reference forwards run in the dev container ------------
@pytest.mark.parametrize("case", ["c2", "c3", "c3w", "c4"])
def test_full_size_pose_and_features_match_the_reference(device, case):
    """BASELINE configs[2..4] against the reference itself at size (oracle/gen_golden.py gen_sized:
    one 20 000-pt 3DMatch-shaped pair; one LiDAR-like 120 000-pt pair pre-voxelised at 0.3 m --
    radial density ~ 1/r, ~2 k superpoints per cloud; eight ModelNet-shaped crops).
    Only summaries are stored: pose, level sizes / widths, statistics and the first rows of the
    conditioned features, overlap scores, match weights, first matches.
    c3w is the KITTI-sized pair with a WELL-conditioned pose solve (synthetic.make_lidar_translated_pair:
    the target is a translated copy of the source, so even randomly initialised features match the true
    correspondences): it is held at north_star's unrelaxed 1e-4.  The conditioning clause below applies
    to the named case c3 only (random matches: the reference's own float32 SVD is the unstable step)."""
    from conftest import load_golden
    from oracle.gen_golden import sized_inputs
    g = load_golden(f"sized_{case}.npz")
    tag, pairs = sized_inputs(case)
    cfg, model = _model(tag, device)
    B = int(g["B"])
    batch = {"src_xyz": [torch.from_numpy(p[0]).to(device) for p in pairs],
             "tgt_xyz": [torch.from_numpy(p[1]).to(device) for p in pairs]}
    with torch.no_grad():
        out = model(batch)
    meta = batch["kpconv_meta"]
    assert [int(p.shape[0]) for p in meta["points"]] == g["level_sizes"].tolist()
    assert [int(meta["neighbors"][l].shape[1]) for l in range(len(meta["points"]))] == g["widths"].tolist()
    for b in range(B):
        err = np.linalg.norm(out["pose"][b].cpu().numpy() - g["pose"][b])
        if not cfg.use_sinkhorn:
            # float64 Kabsch on OUR correspondences / weights: the solve itself must be exact ...
            from oracle import torch_oracle as O
            a64, b64 = out["src_corr"][b].double().cpu(), out["tgt_corr"][b].double().cpu()
            w64 = out["overlap_prob_list"][b].double().cpu()
            T64 = O.compute_rigid_transform(a64, b64, w64).numpy()
            assert np.linalg.norm(out["pose"][b].cpu().numpy() - T64) < 2e-5
            # ... and where the reference (float32 torch.svd) is further from that float64 solution
            # than we are, the 1e-4 bound is applied to the float64 solution instead: with random
            # weights the arg-max matches of the KITTI-scale pair (coordinates to 26 m, weights
            # ~1e-4) give a poorly conditioned covariance
            ref_dev = np.linalg.norm(g["pose"][b] - T64)
            if case == "c3":
                assert err < 1e-4 or (ref_dev > 1e-4 and err <= 1.5 * ref_dev), \
                    f"{case} pair {b}: pose error {err:.2e} (reference vs float64 solve: {ref_dev:.2e})"
            else:
                assert err < 1e-4, f"{case} pair {b}: pose error {err:.2e} (reference vs float64 solve: {ref_dev:.2e})"
        else:
            assert err < 1e-4, f"{case} pair {b}: pose error {err:.2e}"                 # north_star bound
        for side in ("src", "tgt"):
            f = out[f"{side}_feat"][b][0].cpu().numpy()
            st = g[f"{side}_feat_stats{b}"]
            assert f.shape[0] == int(st[3])
            scale = st[2]
            assert np.abs(f[:16] - g[f"{side}_feat_head{b}"]).max() <= 1e-4 * scale      # conditioned features
            assert abs(f.mean() - st[0]) <= 1e-5 * scale and abs(np.abs(f).mean() - st[1]) <= 1e-5 * scale
            ov = out[f"{side}_overlap"][b][0, :64, 0].cpu().numpy()
            assert np.abs(ov - g[f"{side}_overlap_head{b}"]).max() < 1e-4
        v = out["overlap_prob_list"][b].cpu().numpy()
        vs = g[f"val_stats{b}"]
        assert v.shape[0] == int(vs[2]) and abs(v.mean() - vs[0]) <= 5e-3 * vs[0] and abs(v.max() - vs[1]) <= 5e-3 * vs[1]
        ind = out["ind_list"][b][:256].cpu().numpy()
        assert (ind == g[f"ind_head{b}"]).mean() >= 0.99


def test_fp16_attention_mode_end_to_end_on_modelnet_batch256(device):
    """BASELINE configs[4] names "fp16 MFMA attention": spr_set_attn_mode(2) (single-pass fp16 operands,
    fp32 softmax and accumulation) END TO END on the config it exists for, with the accuracy it costs
    stated as bounds:
      * the 8 ModelNet-shaped pairs of golden sized_c4.npz (the reference's own forward at B = 8):
        pose within 2e-3 Frobenius of the REFERENCE, conditioned features within 2e-3 of their scale
        -- i.e. mode 2 does NOT meet north_star's 1e-4 pose bound and is an opt-in throughput mode;
        the default (split) mode on the same pairs stays below 1e-4 (asserted here side by side);
      * the full B = 256 batch against the default mode's poses: median deviation below 5e-4, 90 % of
        the pairs below 5e-3, every pair below 5e-2 (the tail are crops whose weighted Kabsch solve is
        poorly conditioned: a 1e-3 change of the match weights moves them by 1e-2), all rotations proper."""
    from conftest import load_golden
    from oracle.gen_golden import sized_inputs
    cfg, model = _model('modelnet', device)
    g = load_golden("sized_c4.npz")
    _, pairs8 = sized_inputs('c4')
    b8 = {"src_xyz": [torch.from_numpy(p[0]).to(device) for p in pairs8],
          "tgt_xyz": [torch.from_numpy(p[1]).to(device) for p in pairs8]}
    pairs = [synthetic.make_sphere_pair(1024, seed=100 + i) for i in range(256)]
    b256 = {"src_xyz": [torch.from_numpy(p[0]).to(device) for p in pairs],
            "tgt_xyz": [torch.from_numpy(p[1]).to(device) for p in pairs]}
    try:
        with torch.no_grad():
            ops.set_attn_mode(ops.DEFAULT_ATTN_MODE)
            o8_split, o256_split = model(dict(b8)), model(dict(b256))
            ops.set_attn_mode(2)
            o8, o256 = model(dict(b8)), model(dict(b256))
    finally:
        ops.set_attn_mode(ops.DEFAULT_ATTN_MODE)
    worst_pose = worst_feat = worst_split = 0.0
    for b in range(8):
        worst_pose = max(worst_pose, float(np.linalg.norm(o8["pose"][b].cpu().numpy() - g["pose"][b])))
        worst_split = max(worst_split, float(np.linalg.norm(o8_split["pose"][b].cpu().numpy() - g["pose"][b])))
        for side in ("src", "tgt"):
            f = o8[f"{side}_feat"][b][0][:16].cpu().numpy()
            worst_feat = max(worst_feat, float(np.abs(f - g[f"{side}_feat_head{b}"]).max() / g[f"{side}_feat_stats{b}"][2]))
    d256 = (o256["pose"] - o256_split["pose"]).flatten(1).norm(dim=1).cpu().numpy()
    print(f"mode 2 vs reference (8 pairs): pose {worst_pose:.2e}, features {worst_feat:.2e} of scale "
          f"(split mode: pose {worst_split:.2e}); B=256 vs split mode: max {d256.max():.2e}, p99 {np.percentile(d256, 99):.2e}, "
          f"p90 {np.percentile(d256, 90):.2e}, median {np.median(d256):.2e}")
    assert worst_split < 1e-4
    assert worst_pose <= 2e-3 and worst_feat <= 2e-3
    assert np.median(d256) <= 5e-4 and np.percentile(d256, 90) <= 5e-3 and d256.max() <= 5e-2
    _check_poses(o256, 256)
