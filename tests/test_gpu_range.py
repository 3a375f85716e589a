"""GPU: range robustness of the default (split-fp16) arithmetic.

Every matrix product of the library runs, by default, on fp16 hi/lo operand pairs
(three MFMAs per product, fp32 accumulation).  fp16 has a 5-bit exponent, so the
operands are brought into range by per-tensor powers of two derived from measured
(or, for the fused in-projection, derived) magnitude bounds -- spr_common.h
split_pk_s, linear.hip, attention.hip k_plane_scales, kpconv.hip.  These tests
check fp32-level accuracy against float64 at operand magnitudes from 1e-5 to 1e3
(and mixed magnitudes inside one tensor), where an unscaled fp16 split would lose
its low half or overflow.

Criteria (written next to each check):
  * GEMM-shaped ops: per ELEMENT, |err_ij| <= (2^-21 + sqrt(K) 2^-24) sum_k |a_ik b_jk|
    (the forward error bound of an fp32 dot product with 2^-22 operand error), and
    the legacy max-norm criterion 2e-6 max(1, K/256) max|ref|;
  * attention: 3e-6 max|ref|, or 4x the error of a plain fp32 evaluation when the
    scores are so large that fp32 itself is ill-conditioned.
"""
import math

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import torch_oracle as O
from oracle.gen_golden import ops_inputs
from superpoints_registration_amd import get_config, ops, synthetic
from superpoints_registration_amd.regtr import RegTR

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def _check_gemm(got, a64, b64, what, k):
    """got ~ a64 @ b64^T.  Per-element bound relative to sum_k |a||b| + max-norm bound."""
    ref = a64 @ b64.t()
    mag = a64.abs() @ b64.abs().t()
    err = (torch.as_tensor(got, dtype=torch.float64) - ref).abs()
    coef = 2.0 ** -21 + math.sqrt(k) * 2.0 ** -24
    worst = float((err / (mag + 1e-300)).max())
    assert worst <= coef, f"{what}: per-element error {worst:.3e} of sum|a||b| exceeds {coef:.3e}"
    scale = float(ref.abs().max())
    assert float(err.max()) <= 2e-6 * max(1, k // 256) * scale, \
        f"{what}: max err {float(err.max()):.3e} vs scale {scale:.3e}"
    assert torch.isfinite(torch.as_tensor(got)).all(), what


MAGS = [(1e-5, 1e-5), (1e-5, 1e3), (1e-3, 2e-3), (1.0, 2e-4), (3.0, 0.2), (1e3, 1e-5), (1e3, 1e3), (6e4, 1.0)]


@pytest.mark.parametrize("m,k,n", [(700, 256, 256), (300, 1024, 128), (257, 64, 32), (515, 256, 768)])
@pytest.mark.parametrize("mx,mw", MAGS)
def test_linear_every_magnitude(device, m, k, n, mx, mw):
    ops.set_gemm_mode(1)
    x = synthetic.rand((m, k), 11) * mx
    w = synthetic.rand((n, k), 12) * mw
    y = ops.linear(x.to(device), w.to(device)).cpu()
    _check_gemm(y, x.double(), w.double(), f"linear {m}x{k}x{n} |x|~{mx:g} |w|~{mw:g}", k)


def test_linear_mixed_magnitudes_inside_one_tensor(device):
    """Rows of x spanning 4 decades, rows of w spanning 3: the per-tensor scale must not
    destroy the small rows (their absolute error floor is 2^-39 of the tensor maximum)."""
    ops.set_gemm_mode(1)
    m, k, n = 640, 256, 256
    x = synthetic.rand((m, k), 13) * torch.logspace(0, -4, m).unsqueeze(1) * 50.0
    w = synthetic.rand((n, k), 14) * torch.logspace(0, -3, n).unsqueeze(1) * 0.01
    y = ops.linear(x.to(device), w.to(device)).cpu()
    ref = x.double() @ w.double().t()
    mag = x.double().abs() @ w.double().abs().t()
    err = (y.double() - ref).abs()
    assert float((err / mag).max()) <= 2.0 ** -21 + 16 * 2.0 ** -24


def test_linear_bias_residual_activation_small_weights(device):
    ops.set_gemm_mode(1)
    m, k, n = 900, 256, 1024
    x, w = synthetic.rand((m, k), 15, -3, 3), synthetic.rand((n, k), 16) * 2e-3
    b, r = synthetic.rand((n,), 17) * 1e-3, synthetic.rand((m, n), 18) * 1e-3
    y = ops.linear(x.to(device), w.to(device), b.to(device), r.to(device), ops.ACT_RELU).cpu()
    ref = torch.relu(x.double() @ w.double().t() + b.double() + r.double())
    assert float((y.double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())


def _attn_ref(q, k, v, lens, kv_seg, dtype):
    offs = np.concatenate([[0], np.cumsum(lens)])
    out = torch.zeros((sum(lens), 256), dtype=dtype)
    for s in range(len(lens)):
        ks = kv_seg[s]
        qq = q[offs[s]:offs[s + 1]].to(dtype).view(-1, 8, 32).transpose(0, 1)
        kk = k[offs[ks]:offs[ks + 1]].to(dtype).view(-1, 8, 32).transpose(0, 1)
        vv = v[offs[ks]:offs[ks + 1]].to(dtype).view(-1, 8, 32).transpose(0, 1)
        a = torch.softmax(qq @ kk.transpose(1, 2) / math.sqrt(32), -1)
        out[offs[s]:offs[s + 1]] = (a @ vv).transpose(0, 1).reshape(-1, 256)
    return out


@pytest.mark.parametrize("mq,mk,mv", [(1e-5, 1e-5, 1e-5), (1e-3, 1e3, 1e-5), (2.0, 2.0, 1e3), (1e3, 1e-3, 2e-4),
                                      (30.0, 30.0, 1.0), (1e-2, 1e-2, 6e4)])
@pytest.mark.parametrize("mode", [4, 1], ids=["default", "split-everywhere"])
def test_attention_core_every_magnitude(device, mq, mk, mv, mode):
    # mode 4 (default): weights below 2^-5 of their row sum travel in one fp16 plane: <= 3e-5 of the output scale on
    # rows of 30 .. 170 comparable keys (its least accurate case); mode 1: both planes everywhere, fp32 rounding level
    ops.set_attn_mode(mode)
    tol = 3e-5 if mode == 4 else 3e-6
    lens = [170, 33, 129, 1]
    kv_seg = [2, 3, 0, 1]
    tot = sum(lens)
    q = synthetic.rand((tot, 256), 21) * mq
    k = synthetic.rand((tot, 256), 22) * mk
    v = synthetic.rand((tot, 256), 23) * mv
    cu = ops.lengths_to_cu(lens, device)
    seg = torch.tensor(kv_seg, dtype=torch.int32, device=device)
    o = ops.attention(q.to(device), k.to(device), v.to(device), cu, seg, max(lens), 8).cpu()
    ref = _attn_ref(q, k, v, lens, kv_seg, torch.float64)
    ref32 = _attn_ref(q, k, v, lens, kv_seg, torch.float32).double()
    scale = float(ref.abs().max())
    err = float((o.double() - ref).abs().max())
    err32 = float((ref32 - ref).abs().max())
    assert torch.isfinite(o).all()
    ops.set_attn_mode(ops.DEFAULT_ATTN_MODE)
    assert err <= max(tol * scale, 4 * err32), f"attention |q|~{mq:g} |k|~{mk:g} |v|~{mv:g}: {err:.3e} (fp32 {err32:.3e}, scale {scale:.3e})"


@pytest.mark.parametrize("mx,mw,mb", [(1.5, 0.1, 0.2), (1.5, 1e-3, 1e-3), (1e-3, 1e-2, 1e-6), (200.0, 1e-3, 1e-2),
                                      (1e-4, 30.0, 1e-3)])
@pytest.mark.parametrize("shared", [True, False])
@pytest.mark.parametrize("mode", [4, 1], ids=["default", "split-everywhere"])
def test_attention_fused_inprojection_every_magnitude(device, mx, mw, mb, shared, mode):
    ops.set_attn_mode(mode)
    tol = 3e-5 if mode == 4 else 5e-6
    ops.set_gemm_mode(1)
    lens = [301, 70, 257, 33, 129, 1]
    kv_seg = [1, 0, 3, 2, 5, 4]
    tot = sum(lens)
    x_qk = synthetic.rand((tot, 256), 31) * mx
    x_v = x_qk if shared else synthetic.rand((tot, 256), 32) * mx
    w = synthetic.rand((768, 256), 33) * mw
    b = synthetic.rand((768,), 34) * mb
    cu = ops.lengths_to_cu(lens, device)
    seg = torch.tensor(kv_seg, dtype=torch.int32, device=device)
    d_qk = x_qk.to(device)
    d_v = d_qk if shared else x_v.to(device)
    o = ops.attention_inproj(d_qk, d_v, w.to(device), b.to(device), cu, seg, max(lens), 8).cpu()

    def proj(dt):
        qk = x_qk.to(dt) @ w[:512].to(dt).t() + b[:512].to(dt)
        vv = x_v.to(dt) @ w[512:].to(dt).t() + b[512:].to(dt)
        return _attn_ref(qk[:, :256], qk[:, 256:], vv, lens, kv_seg, dt)
    ref, ref32 = proj(torch.float64), proj(torch.float32).double()
    scale = float(ref.abs().max())
    err = float((o.double() - ref).abs().max())
    err32 = float((ref32 - ref).abs().max())
    assert torch.isfinite(o).all()
    ops.set_attn_mode(ops.DEFAULT_ATTN_MODE)
    assert err <= max(tol * scale, 4 * err32), f"fused attention |x|~{mx:g} |w|~{mw:g}: {err:.3e} (fp32 {err32:.3e}, scale {scale:.3e})"


@pytest.mark.parametrize("tag", ["c32", "c64", "c128"])
@pytest.mark.parametrize("mx,mw", [(1e-5, 1e-5), (1.0, 2e-3), (1e3, 1e3), (1e-3, 50.0)])
def test_kpconv_phase2_every_magnitude(device, tag, mx, mw):
    """KPConv: features of magnitude mx (kept non-negative so that the reference's neighbour
    count, #{sum_c x > 0}, is the same for every scale), weights of magnitude mw; float64
    restatement of kpconv_blocks.py:309-412 as the reference."""
    gold, inp = load_golden("ops.npz"), ops_inputs()
    pts = T(inp["kp.pts"])
    nb = T(gold["kp.nb"].astype(np.int64))
    x = inp[f"kp.{tag}.x"].abs() * mx
    w = inp[f"kp.{tag}.w"] / inp[f"kp.{tag}.w"].abs().max() * mw
    kp = T(gold[f"kp.{tag}.kpts"])
    y = ops.kpconv(pts.to(device), pts.to(device), nb.to(torch.int32).to(device), x.to(device), w.to(device),
                   kp.to(device), inp["kp.extent"], rows_sorted=True).cpu()
    ref = O.kpconv(pts.double(), pts.double(), nb, x.double(), w.double(), kp.double(), float(inp["kp.extent"]))
    scale = float(ref.abs().max())
    err = float((y.double() - ref).abs().max())
    assert torch.isfinite(y).all()
    assert err <= 1e-5 * scale, f"kpconv {tag} |x|~{mx:g} |w|~{mw:g}: {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("ms,mt", [(1e-3, 1e3), (1e2, 1e-2), (1.0, 1.0)])
def test_correlation_gemm_unbalanced_features(device, ms, mt):
    """Matching head with src features scaled by ms and tgt features by mt = 1/ms: the
    correlation -- hence the reference's dual-softmax arg-max, values and Sinkhorn weights
    (goldens) -- is unchanged, but the two GEMM operands now sit decades apart."""
    gold, inp = load_golden("ops.npz"), ops_inputs()
    fs, ft = inp["sk.fs"] * ms, inp["sk.ft"] * mt
    feat = torch.cat([fs, ft]).to(device)
    xyz = torch.cat([inp["sk.xs"], inp["sk.xt"]]).to(device)
    cu_host = [0, 60, 107]
    cu = torch.tensor(cu_host, dtype=torch.int32, device=device)
    val, ind = ops.match_dualsoftmax(feat, cu, cu_host, 1)          # N=60 > M=47: lives on tgt tokens
    assert np.array_equal(ind.cpu().numpy()[60:], gold["ds.ind_nm"])
    assert np.abs(val.cpu().numpy()[60:] - gold["ds.val_nm"]).max() <= 1e-5 * np.abs(gold["ds.val_nm"]).max()
    w, that = ops.sinkhorn_correspondences(feat, xyz, cu, cu_host, 1, inp["sk.alpha"], inp["sk.beta"], 3)
    assert np.abs(w.cpu().numpy() - gold["sk.w"]).max() <= 1e-5 * np.abs(gold["sk.w"]).max()


def test_end_to_end_with_small_weights(device):
    """Every weight matrix of the model scaled by 0.01 (trained checkpoints carry weights of
    1e-3..1e-2): pose and conditioned features against the float32 CPU oracle."""
    cfg = get_config("3dmatch")
    src, tgt, _ = synthetic.make_pair(2048, seed=5, extent=0.6, jitter=0.002)
    model = RegTR(cfg)
    synthetic.fill_parameters(model, seed=1)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.dim() >= 2 and not name.endswith(".W"):
                p.mul_(0.01)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(device).eval()
    out = model({"src_xyz": [T(src).to(device)], "tgt_xyz": [T(tgt).to(device)]})
    ref = O.regtr_forward(cfg, sd, [src], [tgt])
    # With weights this small the soft assignment is flat: the soft correspondences t_hat are
    # convex combinations of the target points that differ from one another only in their low bits
    # (spread 2e-3 of the coordinates), and the pose solve amplifies an error in t_hat about
    # 1000 x (scripts/head_err.py: a float32 head's 6e-8 -> 1e-4 in the pose; the reference's own
    # float32 head lands 0.3e-4 .. 1.8e-4 from the float64 pose depending on summation order).
    # The bound is therefore measured: the float64 pose under seeded +-2 ulp(float32)
    # perturbations of t_hat -- what any float32 Sinkhorn head delivers -- with 1e-4 as the floor.
    ref64 = O.regtr_forward(cfg, {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}, [src], [tgt])
    p64 = ref64["pose"][0].numpy().astype(np.float64)
    cs, ct = ref64["cond"][0]
    n_s = int(ref64["lens_c"][0])
    xyz = ref64["xyz_c"].double()
    w64, th64 = O.sinkhorn_soft_correspondences(cs, ct, xyz[n_s:], sd["alpha"], sd["beta"], cfg.sinkhorn_itr)
    gen = torch.Generator().manual_seed(0)
    ulp = torch.tensor(np.spacing(np.abs(th64.numpy()).astype(np.float32)).astype(np.float64))
    noise = 0.0
    for _ in range(8):
        e = (torch.rand(th64.shape, generator=gen, dtype=torch.float64) * 4.0 - 2.0) * ulp
        pn = O.compute_rigid_transform(xyz[:n_s], th64 + e, w64).numpy().astype(np.float64)
        noise = max(noise, float(np.linalg.norm(pn - p64)))
    err = float(np.linalg.norm(out["pose"][0].cpu().numpy().astype(np.float64) - p64))
    assert err < max(1e-4, 2.0 * noise), f"pose error with 0.01x weights: {err:.3e} (float32-head noise: {noise:.3e})"
    sf = out["src_feat"][0][0].cpu().numpy()
    rf = ref64["cond"][0][0].numpy().reshape(sf.shape)
    rn = np.abs(ref["cond"][0][0].numpy().reshape(sf.shape) - rf).max()
    assert np.abs(sf - rf).max() <= max(1e-4 * max(np.abs(rf).max(), 1e-30), 3.0 * rn)


@pytest.mark.parametrize("shared", [True, False])
def test_attention_mode2_single_pass_fp16(device, shared):
    """Mode 2 = hi planes only (11-bit operands, fp32 softmax / accumulation): the throughput
    mode of BASELINE configs[4].  Accuracy target 2e-3 of the output scale."""
    ops.set_attn_mode(2)
    try:
        lens = [301, 70, 257, 33, 129, 1]
        kv_seg = [1, 0, 3, 2, 5, 4]
        tot = sum(lens)
        x_qk = synthetic.rand((tot, 256), 31, -1.5, 1.5)
        x_v = x_qk if shared else synthetic.rand((tot, 256), 32, -1.5, 1.5)
        w, b = synthetic.rand((768, 256), 33, -0.1, 0.1), synthetic.rand((768,), 34, -0.2, 0.2)
        cu = ops.lengths_to_cu(lens, device)
        seg = torch.tensor(kv_seg, dtype=torch.int32, device=device)
        d_qk = x_qk.to(device)
        d_v = d_qk if shared else x_v.to(device)
        o = ops.attention_inproj(d_qk, d_v, w.to(device), b.to(device), cu, seg, max(lens), 8).cpu()
        qk = x_qk.double() @ w[:512].double().t() + b[:512].double()
        vv = x_v.double() @ w[512:].double().t() + b[512:].double()
        ref = _attn_ref(qk[:, :256], qk[:, 256:], vv, lens, kv_seg, torch.float64)
        err = float((o.double() - ref).abs().max())
        assert err <= 2e-3 * float(ref.abs().max()), err
        assert err >= 1e-6 * float(ref.abs().max()), "mode 2 is suspiciously exact: is the hi-only kernel running?"
    finally:
        ops.set_attn_mode(ops.DEFAULT_ATTN_MODE)


def test_attention_mode3_single_probability_plane(device):
    """Mode 3 (round 5) = split-fp16 scores, the probabilities as ONE fp16 plane whose rounded values also form the
    row sum (k_attn_s<true, false>).  Each weight is perturbed by less than 2^-10 relative, the systematic part
    cancels in 1 / l: the error is sum_j w_j delta_j (v_j - o), i.e. at most ~3e-4 of the value spread for a row
    with few effective keys and ~1e-5 for flat rows.  Bound written here: 3e-4 of the output scale on ragged
    segments incl. a 1-token and a 33-token one (few keys: the worst case for this mode)."""
    ops.set_attn_mode(3)
    try:
        lens = [301, 70, 257, 33, 129, 1]
        kv_seg = [1, 0, 3, 2, 5, 4]
        tot = sum(lens)
        g = torch.Generator().manual_seed(11)
        q, k, v = (torch.randn((tot, 256), generator=g) for _ in range(3))
        cu = ops.lengths_to_cu(lens, device)
        seg = torch.tensor(kv_seg, dtype=torch.int32, device=device)
        o = ops.attention(q.to(device), k.to(device), v.to(device), cu, seg, max(lens), 8).cpu()
        ref = _attn_ref(q, k, v, lens, kv_seg, torch.float64)
        err = float((o.double() - ref).abs().max())
        scale = float(ref.abs().max())
        assert err <= 3e-4 * scale, err / scale
        assert err >= 1e-6 * scale, "mode 3 is suspiciously exact: is the single-plane kernel running?"
        ops.set_attn_mode(1)
        o1 = ops.attention(q.to(device), k.to(device), v.to(device), cu, seg, max(lens), 8).cpu()
        assert float((o1.double() - ref).abs().max()) <= 3e-6 * scale
        ops.set_attn_mode(4)
        o4 = ops.attention(q.to(device), k.to(device), v.to(device), cu, seg, max(lens), 8).cpu()
        assert float((o4.double() - ref).abs().max()) <= 3e-5 * scale
    finally:
        ops.set_attn_mode(ops.DEFAULT_ATTN_MODE)


@pytest.mark.parametrize("sharp", [1.0, 4.0, 16.0])
def test_attention_mode4_adaptive_lo_plane(device, sharp):
    """Mode 4 (round 5) = mode 1 with the lo plane of the probabilities only on tiles that hold a probability of at
    least 2^-5 of the lane's running row sum (k_attn_s<..., ADAPT>).  Flat rows (sharp = 1: ~2 000 keys of similar
    weight) run almost entirely on the single-plane path, peaked rows (scores x 4, x 16: a few keys carry the row) get
    their dominant keys with full accuracy: the error stays below 3e-5 of the output scale everywhere, where mode 3
    reaches 1e-4 .. 3e-4 on the peaked cases (asserted: mode 4 must beat mode 3 there)."""
    lens = [1930, 701, 64, 2100]
    kv_seg = [1, 0, 3, 2]
    tot = sum(lens)
    g = torch.Generator().manual_seed(23)
    q = torch.randn((tot, 256), generator=g) * sharp ** 0.5
    k = torch.randn((tot, 256), generator=g) * sharp ** 0.5
    v = torch.randn((tot, 256), generator=g)
    cu = ops.lengths_to_cu(lens, device)
    seg = torch.tensor(kv_seg, dtype=torch.int32, device=device)
    ref = _attn_ref(q, k, v, lens, kv_seg, torch.float64)
    scale = float(ref.abs().max())
    err = {}
    try:
        for mode in (1, 3, 4):
            ops.set_attn_mode(mode)
            o = ops.attention(q.to(device), k.to(device), v.to(device), cu, seg, max(lens), 8).cpu()
            assert torch.isfinite(o).all()
            err[mode] = float((o.double() - ref).abs().max()) / scale
    finally:
        ops.set_attn_mode(ops.DEFAULT_ATTN_MODE)
    assert err[1] <= 6e-6, err        # (scores x 16: the float32 reference itself is at 2e-6 here)
    assert err[4] <= 3e-5, err
    if sharp >= 4.0:
        assert err[4] < 0.5 * err[3], err


@pytest.mark.parametrize("mode", [1, 2, 3, 4])
@pytest.mark.parametrize("pattern", ["rising", "falling", "spike_late", "flat_then_huge"])
def test_attention_deferred_max_recentring(device, mode, pattern):
    """The attention core keeps a per-query reference exponent that is only moved when a later
    key tile would push exp2(s - ref) past fp16's range (attention.hip, "defer-max").  These key
    orders force that rare path: scores rising by > 2^5 from tile to tile, a late spike, and the
    reverse (a huge first tile followed by negligible ones)."""
    ops.set_attn_mode(mode)
    try:
        lens = [700, 333]
        kv_seg = [0, 1]
        tot = sum(lens)
        g = torch.Generator().manual_seed(5)
        q = torch.randn((tot, 256), generator=g)
        k = torch.randn((tot, 256), generator=g) * 0.3
        v = torch.randn((tot, 256), generator=g)
        offs = [0, lens[0], tot]
        for s_ in range(2):
            n = lens[s_]
            j = torch.arange(n, dtype=torch.float32) / n
            if pattern == "rising":
                gain = 0.2 + 6.0 * j                       # scores grow along the key axis
            elif pattern == "falling":
                gain = 6.2 - 6.0 * j
            elif pattern == "spike_late":
                gain = torch.full((n,), 0.3)
                gain[int(0.8 * n)] = 9.0
            else:
                gain = torch.full((n,), 0.05)
                gain[n - 3:] = 12.0
            k[offs[s_]:offs[s_ + 1]] *= gain.unsqueeze(1)
        cu = ops.lengths_to_cu(lens, device)
        seg = torch.tensor(kv_seg, dtype=torch.int32, device=device)
        o = ops.attention(q.to(device), k.to(device), v.to(device), cu, seg, max(lens), 8).cpu()
        ref = _attn_ref(q, k, v, lens, kv_seg, torch.float64)
        ref32 = _attn_ref(q, k, v, lens, kv_seg, torch.float32).double()
        scale = float(ref.abs().max())
        err = float((o.double() - ref).abs().max())
        err32 = float((ref32 - ref).abs().max())
        tol = {1: 3e-6, 2: 3e-3, 3: 3e-4, 4: 3e-5}[mode]
        assert torch.isfinite(o).all()
        assert err <= max(tol * scale, 4 * err32), f"{pattern} mode {mode}: {err:.3e} (fp32 {err32:.3e}, scale {scale:.3e})"
    finally:
        ops.set_attn_mode(ops.DEFAULT_ATTN_MODE)


def test_published_ranges_replace_the_measuring_pass(device):
    """Operand-range hand-over (spr.h): LayerNorm, the ReLU GEMM and the fused attention publish
    the range of what they wrote, and the consuming GEMM must produce the same result as when it
    measures the operand itself -- including for inputs far from unit scale."""
    ops.set_gemm_mode(1)
    ops.set_attn_mode(ops.DEFAULT_ATTN_MODE)
    for mag in (1.0, 1e-4, 3e3):
        x = synthetic.rand((700, 256), 41, -2, 3) * mag
        g, b = synthetic.rand((256,), 42, 0.5, 1.5) * mag, synthetic.rand((256,), 43) * mag
        pos = synthetic.rand((700, 256), 44) * mag
        w1, b1 = synthetic.rand((1024, 256), 45, -0.1, 0.1), synthetic.rand((1024,), 46) * mag
        w2 = synthetic.rand((256, 1024), 47, -0.1, 0.1)
        n, npos = ops.layernorm(x.to(device), g.to(device), b.to(device), 1e-5, pos=pos.to(device))
        assert getattr(n, "_spr_range", None) is not None and getattr(npos, "_spr_range", None) is not None
        h = ops.linear(n, w1.to(device), b1.to(device), act=ops.ACT_RELU)          # consumes n's range, publishes h's
        assert ops._get_range(h)[1] > 0
        y = ops.linear(h, w2.to(device))                                           # consumes h's range
        # the same chain with every range dropped (clones are new tensors without the attribute)
        h2 = ops.linear(n.clone(), w1.to(device), b1.to(device), act=ops.ACT_RELU)
        y2 = ops.linear(h2.clone(), w2.to(device))
        assert torch.equal(h, h2), mag           # same power-of-two scales -> bitwise the same GEMM
        ref = torch.relu(n.double().cpu() @ w1.double().t() + b1.double()) @ w2.double().t()
        assert float((y.double().cpu() - ref).abs().max()) <= 4e-6 * float(ref.abs().max())
        assert float((y2.double().cpu() - ref).abs().max()) <= 4e-6 * float(ref.abs().max())
    # a stale range must not be used: in-place modification bumps the version counter
    n, _ = ops.layernorm(x.to(device), g.to(device), b.to(device), 1e-5)
    n.mul_(1000.0)
    assert ops._get_range(n) == (None, 0)
    y = ops.linear(n, w1.to(device))
    ref = n.double().cpu() @ w1.double().t()
    assert torch.isfinite(y).all() and float((y.double().cpu() - ref).abs().max()) <= 4e-6 * float(ref.abs().max())


def test_weight_ranges_are_cached_per_version_and_follow_in_place_updates(device):
    """ops.linear keeps the measured range of a weight tensor on the tensor (no pre-pass on later
    calls); an in-place update moves the version counter and must trigger a new measurement --
    otherwise a grown weight would overflow the fp16 planes."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(300, 64, generator=g).to(device)
    w = (torch.randn(96, 64, generator=g) * 0.05).to(device)
    y0 = ops.linear(x, w)
    assert getattr(w, '_spr_range', None) is not None
    first = w._spr_range[0]
    ops.linear(x, w)
    assert w._spr_range[0] is first                       # cached: same partials object
    with torch.no_grad():
        w.mul_(4096.0)                                    # far beyond the one bit of headroom
    y1 = ops.linear(x, w)
    assert w._spr_range[0] is not first
    ref = x.double() @ w.double().t()
    assert torch.isfinite(y1).all()
    assert (y1.double() - ref).abs().max() <= 2e-6 * ref.abs().max()
    assert (y0.double() * 4096.0 - ref).abs().max() <= 4e-6 * ref.abs().max()
    ops.invalidate_ranges()
    ops.linear(x, w)
    assert w._spr_range[4] == ops._range_epoch[0]


def test_prime_weight_ranges_measures_all_stale_weights_in_one_launch(device):
    """ops.prime_weight_ranges (spr_absmax_multi): the ranges attached in one launch are upper bounds equal to the
    true maxima, are picked up by _static_range without another measurement, and go stale with the version counter."""
    g = torch.Generator().manual_seed(2)
    ws = [torch.randn(shape, generator=g).mul(scale).to(device).requires_grad_(True)
          for shape, scale in (((256, 256), 1.0), ((1024, 256), 1e-3), ((15, 32, 32), 40.0), ((3, 5), 7.0), ((768, 256), 1e4))]
    bias = torch.randn(256, generator=g).to(device)                     # 1-D: not a product operand, skipped
    assert ops.prime_weight_ranges(ws + [bias]) == len(ws)
    for w in ws:
        parts, n = ops._get_range(w)
        assert parts is not None and n == 16
        assert float(parts[:n].max()) == float(w.detach().abs().max())
        again, n2 = ops._static_range(w)
        assert again.data_ptr() == parts.data_ptr() and n2 == n            # no second measurement
    assert ops.prime_weight_ranges(ws) == 0                             # nothing stale
    with torch.no_grad():
        ws[1].mul_(3.0)                                                  # an optimizer step bumps the version
    assert ops._get_range(ws[1])[0] is None
    assert ops.prime_weight_ranges(ws) == 1
    assert float(ops._get_range(ws[1])[0].max()) == float(ws[1].detach().abs().max())
