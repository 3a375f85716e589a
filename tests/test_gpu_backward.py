"""GPU: the explicit HIP backward (autograd.py over the C ABI's backward entry points) against
  * torch autograd of the float64 CPU oracle, operator by operator, and
  * the REFERENCE's own parameter gradients (`compute_loss(model(batch), batch)[...].backward()`,
    golden fixtures tests/golden/grad_*_b2.npz made by oracle/gen_golden.py gen_grad) on the three
    shipped configs -- for the partial loss 0.1 feature + overlap (encoder + transformer + loss
    kernels) and for the total loss (adds the pose head: Sinkhorn / dual-softmax, Kabsch).
Tolerance: 1e-4 relative per parameter tensor (|g - g_ref| <= 1e-4 ||g_ref||_inf-scale on the pinned
entries, norms within 1e-4 relative), written next to each check.
"""
import math

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import torch_oracle as O
from oracle.gen_golden import grad_sample_indices, loss_inputs, ops_inputs, pairs_for
from superpoints_registration_amd import get_config, ops, synthetic
from superpoints_registration_amd.regtr import RegTR

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def _rel(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return float((got - ref).abs().max() / max(float(ref.abs().max()), 1e-30))


def _leaf(t, device):
    return t.clone().to(device).requires_grad_(True)


def test_linear_backward(device):
    m, k, n = 777, 256, 192
    x, w, b, r = synthetic.rand((m, k), 1, -2, 2), synthetic.rand((n, k), 2, -0.2, 0.2), synthetic.rand((n,), 3), \
        synthetic.rand((m, n), 4)
    go = synthetic.rand((m, n), 5)
    for act, f in ((ops.ACT_NONE, lambda t: t), (ops.ACT_RELU, torch.relu), (ops.ACT_SIGMOID, torch.sigmoid)):
        dx, dw, db, dr = _leaf(x, device), _leaf(w, device), _leaf(b, device), _leaf(r, device)
        ops.linear(dx, dw, db, dr, act).backward(go.to(device))
        cx, cw, cb, cr = (t.double().requires_grad_(True) for t in (x, w, b, r))
        f(cx @ cw.t() + cb + cr).backward(go.double())
        for g, c, nm in ((dx, cx, "dx"), (dw, cw, "dw"), (db, cb, "db"), (dr, cr, "dres")):
            assert _rel(g.grad, c.grad) <= 2e-5, f"linear act {act} {nm}: {_rel(g.grad, c.grad):.2e}"


def test_layernorm_backward(device):
    x, g, b, p = synthetic.rand((321, 256), 5, -3, 5), synthetic.rand((256,), 6, 0.5, 1.5), \
        synthetic.rand((256,), 7), synthetic.rand((321, 256), 8)
    g1, g2 = synthetic.rand((321, 256), 9), synthetic.rand((321, 256), 10)
    dx, dg, db = _leaf(x, device), _leaf(g, device), _leaf(b, device)
    n, npos = ops.layernorm(dx, dg, db, 1e-5, pos=p.to(device))
    (n * g1.to(device)).sum().add((npos * g2.to(device)).sum()).backward()
    cx, cg, cb = (t.double().requires_grad_(True) for t in (x, g, b))
    y = torch.nn.functional.layer_norm(cx, (256,), cg, cb, 1e-5)
    ((y * g1.double()).sum() + ((y + p.double()) * g2.double()).sum()).backward()
    for a, c, nm in ((dx, cx, "dx"), (dg, cg, "dgamma"), (db, cb, "dbeta")):
        assert _rel(a.grad, c.grad) <= 2e-5, f"layernorm {nm}: {_rel(a.grad, c.grad):.2e}"


def test_instnorm_lrelu_add_backward(device):
    inp = ops_inputs()
    lens = inp["kp.lens"]
    cu = ops.lengths_to_cu(lens.tolist(), device)
    x, add, go = inp["in.x"], synthetic.rand((600, 64), 77), synthetic.rand((600, 64), 78)
    dx, da = _leaf(x, device), _leaf(add, device)
    ops.instnorm(dx, cu, add=da, slope=0.1).backward(go.to(device))
    cx, ca = x.double().requires_grad_(True), add.double().requires_grad_(True)
    torch.nn.functional.leaky_relu(O.instance_norm(cx, lens) + ca, 0.1).backward(go.double())
    assert _rel(dx.grad, cx.grad) <= 2e-5 and _rel(da.grad, ca.grad) <= 1e-6
    # without normalisation (bias-only BatchNormBlock variant)
    dx2 = _leaf(x, device)
    ops.instnorm(dx2, cu, norm=False, slope=0.1).backward(go.to(device))
    cx2 = x.double().requires_grad_(True)
    torch.nn.functional.leaky_relu(cx2, 0.1).backward(go.double())
    assert _rel(dx2.grad, cx2.grad) <= 1e-6


def test_maxpool_and_gather_backward(device):
    gold, inp = load_golden("ops.npz"), ops_inputs()
    idx = T(gold["mp.idx"].astype(np.int64))
    x, go = inp["in.x"], synthetic.rand((idx.shape[0], 64), 80)
    dx = _leaf(x, device)
    ops.maxpool(dx, idx.to(torch.int32).to(device)).backward(go.to(device))
    cx = x.double().requires_grad_(True)
    O.max_pool(cx, idx).backward(go.double())
    assert _rel(dx.grad, cx.grad) <= 1e-6
    sel = torch.tensor([5, 0, 5, 599, 17], dtype=torch.int32)
    dx2 = _leaf(x, device)
    ops.gather_rows(dx2, sel.to(device)).backward(go[:5].to(device))
    cx2 = x.double().requires_grad_(True)
    cx2[sel.long()].backward(go[:5].double())
    assert _rel(dx2.grad, cx2.grad) <= 1e-6
    # a NaN / inf in the incoming gradient must not come out finite (the fixed-point sums cannot
    # carry it per element, so the whole gradient is marked: a finite-gradient check downstream still fires)
    for bad in (float('nan'), float('inf')):
        gb = go.clone()
        gb[3, 7] = bad
        dx3 = _leaf(x, device)
        ops.maxpool(dx3, idx.to(torch.int32).to(device)).backward(gb.to(device))
        assert not bool(torch.isfinite(dx3.grad).all())
        dx4 = _leaf(x, device)
        ops.gather_rows(dx4, sel.to(device)).backward(gb[:5].to(device))
        assert not bool(torch.isfinite(dx4.grad).all())


@pytest.mark.parametrize("tag", ["c32", "c64", "c128", "c48"])
def test_kpconv_backward(device, tag):
    gold, inp = load_golden("ops.npz"), ops_inputs()
    pts, nb = T(inp["kp.pts"]), T(gold["kp.nb"].astype(np.int64))
    x, w, kp = inp[f"kp.{tag}.x"], inp[f"kp.{tag}.w"], T(gold[f"kp.{tag}.kpts"])
    go = synthetic.rand((600, w.shape[2]), 90)
    dx, dw = _leaf(x, device), _leaf(w, device)
    ops.kpconv(pts.to(device), pts.to(device), nb.to(torch.int32).to(device), dx, dw, kp.to(device), inp["kp.extent"],
               rows_sorted=True).backward(go.to(device))
    cx, cw = x.double().requires_grad_(True), w.double().requires_grad_(True)
    O.kpconv(pts.double(), pts.double(), nb, cx, cw, kp.double(), float(inp["kp.extent"])).backward(go.double())
    assert _rel(dx.grad, cx.grad) <= 2e-5, f"kpconv {tag} dx {_rel(dx.grad, cx.grad):.2e}"
    assert _rel(dw.grad, cw.grad) <= 2e-5, f"kpconv {tag} dW {_rel(dw.grad, cw.grad):.2e}"


@pytest.fixture(params=[4, 1, 0], ids=["default", "split-fp16", "exact-f32"])
def attn_mode(request):
    """Both arithmetic forms of the attention kernels (forward AND backward follow spr_set_attn_mode)."""
    ops.set_attn_mode(request.param)
    yield request.param
    ops.set_attn_mode(ops.DEFAULT_ATTN_MODE)


def test_attention_backward(device, attn_mode):
    lens, kv_seg = [170, 33, 129, 65], [2, 3, 0, 1]
    tot = sum(lens)
    qkv = synthetic.rand((tot, 768), 9, -1.5, 1.5)
    go = synthetic.rand((tot, 256), 11)
    dq = _leaf(qkv, device)
    cu = ops.lengths_to_cu(lens, device)
    seg = torch.tensor(kv_seg, dtype=torch.int32, device=device)
    ops.attention(dq[:, :256], dq[:, 256:512], dq[:, 512:], cu, seg, max(lens), 8, lens_host=lens,
                  kv_seg_host=kv_seg).backward(go.to(device))
    cq = qkv.double().requires_grad_(True)
    offs = np.concatenate([[0], np.cumsum(lens)])
    outs = []
    for s in range(4):
        ks = kv_seg[s]
        q = cq[offs[s]:offs[s + 1], :256].view(-1, 8, 32).transpose(0, 1)
        k = cq[offs[ks]:offs[ks + 1], 256:512].view(-1, 8, 32).transpose(0, 1)
        v = cq[offs[ks]:offs[ks + 1], 512:].view(-1, 8, 32).transpose(0, 1)
        a = torch.softmax(q @ k.transpose(1, 2) / math.sqrt(32), -1)
        outs.append((a @ v).transpose(0, 1).reshape(-1, 256))
    torch.cat(outs).backward(go.double())
    assert _rel(dq.grad, cq.grad) <= 2e-5, f"attention dqkv {_rel(dq.grad, cq.grad):.2e}"


def _attention_f64(q0, k0, v0, go, lens, kv_seg):
    cq, ck, cv = (t.double().requires_grad_(True) for t in (q0, k0, v0))
    offs = np.concatenate([[0], np.cumsum(lens)])
    outs = []
    for s in range(len(lens)):
        ks = kv_seg[s]
        q = cq[offs[s]:offs[s + 1]].view(-1, 8, 32).transpose(0, 1)
        k = ck[offs[ks]:offs[ks + 1]].view(-1, 8, 32).transpose(0, 1)
        v = cv[offs[ks]:offs[ks + 1]].view(-1, 8, 32).transpose(0, 1)
        a = torch.softmax(q @ k.transpose(1, 2) / math.sqrt(32), -1)
        outs.append((a @ v).transpose(0, 1).reshape(-1, 256))
    torch.cat(outs).backward(go.double())
    return cq.grad, ck.grad, cv.grad


@pytest.mark.parametrize("qs,ks,vs,gs", [(1.0, 1.0, 1.0, 1.0), (1e-4, 3e3, 1e5, 1e-7), (40.0, 1.0, 1e-6, 1e6),
                                         (6.0, 6.0, 1.0, 1.0)])
def test_attention_backward_split_operand_ranges(device, qs, ks, vs, gs):
    """The split-fp16 backward scales every operand by a measured power of two and dS by a static bound: operands
    far from 1, and peaked softmax rows (|s| ~ 40: P ~ 1, the largest dS), keep the float32-level accuracy."""
    lens, kv_seg = [200, 150, 70, 131], [1, 0, 3, 2]
    tot = sum(lens)
    q0, k0, v0 = (synthetic.rand((tot, 256), s, -1.0, 1.0) * m for s, m in ((31, qs), (32, ks), (33, vs)))
    go = synthetic.rand((tot, 256), 34) * gs
    cu = ops.lengths_to_cu(lens, device)
    seg = torch.tensor(kv_seg, dtype=torch.int32, device=device)
    lq, lk, lv = _leaf(q0, device), _leaf(k0, device), _leaf(v0, device)
    ops.attention(lq, lk, lv, cu, seg, max(lens), 8, lens_host=lens, kv_seg_host=kv_seg).backward(go.to(device))
    for got, ref, nm in zip((lq.grad, lk.grad, lv.grad), _attention_f64(q0, k0, v0, go, lens, kv_seg), "qkv"):
        assert torch.isfinite(got).all()
        assert _rel(got, ref) <= 2e-5, f"attention d{nm} at scales {(qs, ks, vs, gs)}: {_rel(got, ref):.2e}"


def test_attention_backward_operand_planes_and_small_workspace_agree(device, monkeypatch):
    """spr_attn_varlen_bwd writes the split operand planes once per call when the workspace has room for them
    (spr_attn_bwd_workspace_bytes2) and lets its kernels convert the fp32 tiles themselves when it has not
    (spr_attn_bwd_workspace_bytes): the same values reach the same LDS tiles, so the gradients agree bit for bit."""
    from superpoints_registration_amd import _lib
    lens, kv_seg = [130, 64, 1, 257], [1, 0, 3, 2]
    tot = sum(lens)
    q0, k0, v0 = (synthetic.rand((tot, 256), s, -1.5, 1.5) for s in (41, 42, 43))
    go = synthetic.rand((tot, 256), 44)
    cu = ops.lengths_to_cu(lens, device)
    seg = torch.tensor(kv_seg, dtype=torch.int32, device=device)

    def grads():
        lq, lk, lv = _leaf(q0, device), _leaf(k0, device), _leaf(v0, device)
        ops.attention(lq, lk, lv, cu, seg, max(lens), 8, lens_host=lens, kv_seg_host=kv_seg).backward(go.to(device))
        return lq.grad.clone(), lk.grad.clone(), lv.grad.clone()

    with_planes = grads()
    L = _lib.lib()
    small = L.spr_attn_bwd_workspace_bytes

    class Patched:
        def __getattr__(self, name):
            if name == "spr_attn_bwd_workspace_bytes2":
                return lambda t, nseg, nhead: small(t, nhead)
            return getattr(L, name)

    monkeypatch.setattr(ops, "_workspace", lambda n, dev: torch.empty(max(int(n), 1), dtype=torch.uint8, device=dev))
    monkeypatch.setattr(_lib, "lib", lambda: Patched())
    without = grads()
    for a, b, nm in zip(with_planes, without, "qkv"):
        assert torch.equal(a, b), f"d{nm}"
    for got, ref, nm in zip(with_planes, _attention_f64(q0, k0, v0, go, lens, kv_seg), "qkv"):
        assert _rel(got, ref) <= 2e-5, f"attention d{nm} {_rel(got, ref):.2e}"


def test_attention_backward_many_tiles_and_reproducible(device, attn_mode):
    """Several 64-row tiles per segment on both sides, self and cross segments of different lengths; two runs
    must agree bit for bit (fixed summation order: no atomics in spr_attn_varlen_bwd)."""
    lens, kv_seg = [450, 321, 64, 577], [1, 0, 3, 2]
    tot = sum(lens)
    q0, k0, v0 = (synthetic.rand((tot, 256), s, -2.0, 2.0) for s in (21, 22, 23))
    go = synthetic.rand((tot, 256), 24)
    cu = ops.lengths_to_cu(lens, device)
    seg = torch.tensor(kv_seg, dtype=torch.int32, device=device)
    grads = []
    for _ in range(2):
        lq, lk, lv = _leaf(q0, device), _leaf(k0, device), _leaf(v0, device)
        ops.attention(lq, lk, lv, cu, seg, max(lens), 8, lens_host=lens, kv_seg_host=kv_seg).backward(go.to(device))
        grads.append((lq.grad.clone(), lk.grad.clone(), lv.grad.clone()))
    for a, b in zip(grads[0], grads[1]):
        assert torch.equal(a, b)
    for got, ref, nm in zip(grads[0], _attention_f64(q0, k0, v0, go, lens, kv_seg), "qkv"):
        assert _rel(got, ref) <= 2e-5, f"attention d{nm} {_rel(got, ref):.2e}"      # both forms, fp32 softmax


def test_procrustes_backward(device):
    inp = ops_inputs()
    a, b, w = inp["rt.a"].reshape(-1, 3), inp["rt.b"].reshape(-1, 3), inp["rt.w"].reshape(-1)
    pair_cu = torch.tensor([0, 200, 400, 600], dtype=torch.int32, device=device)
    go = synthetic.rand((3, 3, 4), 12)
    db, dw = _leaf(b, device), _leaf(w, device)
    ops.weighted_procrustes(a.to(device), db, dw, pair_cu).backward(go.to(device))
    cb, cw = b.double().requires_grad_(True), w.double().requires_grad_(True)
    poses = [O.compute_rigid_transform(a.double()[200 * k:200 * (k + 1)], cb[200 * k:200 * (k + 1)], cw[200 * k:200 * (k + 1)])
             for k in range(3)]
    torch.stack(poses).backward(go.double())
    for k in range(3):   # the mirrored / nearly planar set (k = 1) is ill-conditioned: looser there
        tol = 2e-5 if k != 1 else 2e-3
        sl = slice(200 * k, 200 * (k + 1))
        assert _rel(db.grad[sl], cb.grad[sl]) <= tol, f"procrustes db set {k}: {_rel(db.grad[sl], cb.grad[sl]):.2e}"
        assert _rel(dw.grad[sl], cw.grad[sl]) <= tol, f"procrustes dw set {k}: {_rel(dw.grad[sl], cw.grad[sl]):.2e}"


def test_sinkhorn_and_match_backward(device):
    inp = ops_inputs()
    fs, ft, xs, xt = inp["sk.fs"], inp["sk.ft"], inp["sk.xs"], inp["sk.xt"]
    feat = torch.cat([fs, ft])
    xyz = torch.cat([xs, xt]).to(device)
    cu_host = [0, 60, 107]
    cu = torch.tensor(cu_host, dtype=torch.int32, device=device)
    gw, gt = synthetic.rand((60,), 13), synthetic.rand((60, 3), 14)
    df = _leaf(feat, device)
    al, be = _leaf(torch.tensor(0.9), device), _leaf(torch.tensor(1.1), device)
    w, that = ops.sinkhorn_correspondences(df, xyz, cu, cu_host, 1, al, be, 3)
    ((w * gw.to(device)).sum() + (that * gt.to(device)).sum()).backward()
    cf = feat.double().requires_grad_(True)
    ca, cb = torch.tensor(0.9, dtype=torch.float64, requires_grad=True), torch.tensor(1.1, dtype=torch.float64, requires_grad=True)
    score = torch.clamp(cf[:60] @ cf[60:].t() / 16.0, min=0.0)
    aff = -(score - torch.nn.functional.softplus(ca)) / (torch.exp(cb) + 0.02)
    u = torch.zeros(60, dtype=torch.float64)
    v = torch.zeros(47, dtype=torch.float64)
    for _ in range(3):
        u = torch.log1p(torch.exp(aff - v[None, :]).sum(1))
        v = torch.log1p(torch.exp(aff - u[:, None]).sum(0))
    P = torch.exp(aff - u[:, None] - v[None, :])
    wr = P.sum(1)
    tr = P @ xt.double() / (wr[:, None] + 1e-6)
    ((wr * gw.double()).sum() + (tr * gt.double()).sum()).backward()
    assert _rel(df.grad, cf.grad) <= 5e-5, f"sinkhorn dfeat {_rel(df.grad, cf.grad):.2e}"
    assert abs(float(al.grad) - float(ca.grad)) <= 5e-5 * abs(float(ca.grad))
    assert abs(float(be.grad) - float(cb.grad)) <= 5e-5 * abs(float(cb.grad))
    # dual-softmax values
    gv = synthetic.rand((107,), 15)
    df2 = _leaf(feat, device)
    val, ind = ops.match_dualsoftmax(df2, cu, cu_host, 1)
    (val[60:] * gv[60:].to(device)).sum().backward()
    cf2 = feat.double().requires_grad_(True)
    corr = cf2[:60] @ cf2[60:].t() / 16.0
    attn = torch.softmax(corr, 0) * torch.softmax(corr, 1)
    vr, _ = attn.max(0)
    (vr * gv[60:].double()).sum().backward()
    assert _rel(df2.grad, cf2.grad) <= 5e-5, f"dual softmax dfeat {_rel(df2.grad, cf2.grad):.2e}"


def test_loss_kernels_backward(device):
    inp = ops_inputs()
    fs, ft, xs, xt = inp["sk.fs"], inp["sk.ft"], inp["sk.xs"] * 0.3, inp["sk.xt"] * 0.3
    W = synthetic.rand((256, 256), 16, -0.05, 0.05)
    pose = torch.tensor([[1.0, 0, 0, 0.02], [0, 1, 0, -0.01], [0, 0, 1, 0.0]])
    da, dp, dW = _leaf(fs, device), _leaf(ft, device), _leaf(W, device)
    ops.infonce_pair(da, dp, xs.to(device), pose.to(device), xt.to(device), dW, 0.2, 0.4).backward()
    ca, cp, cW = (t.double().requires_grad_(True) for t in (fs, ft, W))
    ax = O.se3_transform(pose.double(), xs.double())
    O.infonce(ca, cp, ax, xt.double(), cW, 0.2, 0.4).backward()
    for g, c, nm in ((da, ca, "anchor"), (dp, cp, "positive"), (dW, cW, "W")):
        assert _rel(g.grad, c.grad) <= 5e-5, f"infonce {nm}: {_rel(g.grad, c.grad):.2e}"
    x, y = synthetic.rand((500,), 17, 0.01, 0.99), (synthetic.rand((500,), 18) > 0).float()
    dx = _leaf(x, device)
    ops.bce_logits_mean(dx, y.to(device)).backward()
    cx = x.double().requires_grad_(True)
    torch.nn.functional.binary_cross_entropy_with_logits(cx, y.double()).backward()
    assert _rel(dx.grad, cx.grad) <= 1e-5
    pp = (pose + 0.05 * synthetic.rand((3, 4), 19))
    dpp = _leaf(pp, device)
    ops.transform_l1_pair(pose.to(device), dpp, xs.to(device)).backward()
    cpp = pp.double().requires_grad_(True)
    (O.se3_transform(pose.double(), xs.double()) - O.se3_transform(cpp, xs.double())).abs().mean().backward()
    assert _rel(dpp.grad, cpp.grad) <= 1e-5


def _train_step(tag, device, which):
    g = load_golden(f"grad_{tag}_b2.npz")
    B = int(g["B"])
    cfg = get_config(tag)
    pairs, sizes = pairs_for(tag, B)
    pose, src_ov, tgt_ov = loss_inputs(tag, B)
    model = RegTR(cfg)
    synthetic.fill_parameters(model, seed=int(g["seed"]))
    model = model.to(device).train()
    batch = {"src_xyz": [T(p[0][:n]).to(device) for p, (n, m) in zip(pairs, sizes)],
             "tgt_xyz": [T(p[1][:m]).to(device) for p, (n, m) in zip(pairs, sizes)],
             "pose": T(pose).to(device),
             "src_overlap": [T(o).to(device) for o in src_ov], "tgt_overlap": [T(o).to(device) for o in tgt_ov]}
    out = model(batch)
    losses = model.compute_loss(out, batch)
    loss = losses["total"] if which == "total" else 0.1 * losses["feature"] + losses["overlap"]
    model.zero_grad(set_to_none=True)
    loss.backward()
    return g, model, losses


_F64_SCALARS = {}


def _f64_sinkhorn_scalar_grads(tag, which):
    """d loss / d alpha, d beta of the same training step by the float64 CPU oracle's autograd."""
    key = (tag, which)
    if key not in _F64_SCALARS:
        from oracle import torch_oracle as O
        g = load_golden(f"grad_{tag}_b2.npz")
        B = int(g["B"])
        cfg = get_config(tag)
        pairs, sizes = pairs_for(tag, B)
        pose, src_ov, tgt_ov = loss_inputs(tag, B)
        model = RegTR(cfg)
        synthetic.fill_parameters(model, seed=int(g["seed"]))
        sd = {k: (v.detach().clone().double() if v.is_floating_point() else v.clone()) for k, v in model.state_dict().items()}
        for k in ("alpha", "beta"):
            sd[k].requires_grad_(True)
        fwd = O.regtr_forward(cfg, sd, [p[0][:n] for p, (n, m) in zip(pairs, sizes)],
                              [p[1][:m] for p, (n, m) in zip(pairs, sizes)])
        L = O.compute_loss(cfg, sd, fwd, pose, src_ov, tgt_ov)
        (L["total"] if which == "total" else 0.1 * L["feature"] + L["overlap"]).backward()
        _F64_SCALARS[key] = {k: float(sd[k].grad) for k in ("alpha", "beta")}
    return _F64_SCALARS[key]


def test_backward_is_bitwise_reproducible(device):
    """Every sum of the backward has a fixed order -- deterministic split-K partials, the flash-style attention
    backward, and 64-bit fixed-point integer atomics for the three scatter-adds (max-pool, row gather, KPConv
    neighbour gather: float atomics before round 3) -- so two runs of the same step give the same bits in
    every parameter gradient."""
    _, m1, _ = _train_step("3dmatch", device, "total")
    _, m2, _ = _train_step("3dmatch", device, "total")
    n = 0
    for (k1, p1), (k2, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        assert k1 == k2
        if p1.grad is None:
            assert p2.grad is None
            continue
        assert torch.equal(p1.grad, p2.grad), k1
        n += 1
    assert n > 100


@pytest.mark.parametrize("which", ["fo", "total"])
@pytest.mark.parametrize("tag", ["3dmatch", "kitti", "modelnet"])
def test_parameter_gradients_match_the_reference(device, tag, which):
    """One training step's parameter gradients against the reference's own backward.

    Criteria per parameter tensor (g = ours, r = reference):
      * outside the KPConv encoder (transformer, projections, loss parameters):
        | ||g|| - ||r|| | <= 1e-4 ||r|| and every pinned entry within 1e-4 of the tensor's scale;
      * KPConv-encoder tensors and the two Sinkhorn scalars alpha / beta (sums over all N x M
        affinities): norm within 5e-4 (1e-3 for the scalars), RMS deviation of the pinned entries
        <= 2e-2 of their RMS; at most half of the encoder tensors may exceed the 1e-4 entry bound.
    Why the encoder is different: its gradients pass through LeakyReLU / max-pool / K-nearest
    DECISIONS.  Our forward agrees with the reference to ~1e-6; an activation that close to zero
    takes the other LeakyReLU branch, which changes one element of one upstream gradient by a
    factor 10 and hence one row of a weight gradient (a sum of a few hundred signed terms) by a
    few per cent -- tests/test_backward_conditioning.py shows the same jump inside the float64
    oracle itself.  Every operator's backward is checked separately
    at 2e-5 above, where both sides see identical inputs, and the whole encoder entry by entry at
    2e-5 with the decisions frozen in test_encoder_gradients_entrywise_with_frozen_decisions below --
    the norm / RMS window here only covers what is left: the effect of the flips themselves."""
    g, model, losses = _train_step(tag, device, which)
    for k in ("feature", "T", "overlap", "total"):
        assert abs(float(losses[k].detach()) - float(g[f"loss_{k}"])) <= 5e-5 * max(1.0, abs(float(g[f"loss_{k}"])))
    n_checked, n_enc, enc_loose, report = 0, 0, 0, []
    for name, p in model.named_parameters():
        if f"{which}|{name}|none" in g:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, f"{name}: the reference has no gradient here"
            continue
        if not p.requires_grad:
            continue
        assert p.grad is not None, f"{name}: no gradient"
        gr = p.grad.detach().double().reshape(-1).cpu().numpy()
        ref_norm = float(g[f"{which}|{name}|norm"])
        if f"{which}|{name}|full" in g:
            ref_e, got_e = g[f"{which}|{name}|full"].astype(np.float64), gr
        else:
            ref_e = g[f"{which}|{name}|samples"].astype(np.float64)
            got_e = gr[grad_sample_indices(name, gr.size)]
        scale = max(np.abs(ref_e).max(), ref_norm / math.sqrt(gr.size), 1e-30)
        err = np.abs(got_e - ref_e).max() / scale
        rms = np.sqrt(np.mean((got_e - ref_e) ** 2)) / max(np.sqrt(np.mean(ref_e ** 2)), 1e-30)
        nerr = abs(np.linalg.norm(gr) - ref_norm) / max(ref_norm, 1e-30)
        n_checked += 1
        if name in ("alpha", "beta"):
            # Sums over all N x M affinities with heavy cancellation: at fp32 the reference's OWN value sits 3e-4 (alpha) /
            # 5e-5 (beta) from the float64 oracle's autograd on the same step (scripts/dalpha_probe.py), so 1e-4
            # against the reference is not a meaningful bar.  Anchor = the float64 oracle: ours must be within 1e-4 of
            # it, or at most twice as far from it as the reference's fp32 backward is (observed: 3.9e-4 vs 2.7e-4 for
            # alpha, 2.9e-5 vs 4.7e-5 for beta); the window against the reference itself stays at 1e-3.
            assert nerr <= 1e-3, f"{tag}/{which} {name}: deviates by {nerr:.2e}"
            f64 = _f64_sinkhorn_scalar_grads(tag, which)[name]
            ours, ref_v = float(gr[0]), float(ref_e.reshape(-1)[0])
            e_ours, e_ref = abs(ours - f64) / abs(f64), abs(ref_v - f64) / abs(f64)
            assert e_ours <= max(1e-4, 2.0 * e_ref), f"{tag}/{which} d{name}: {e_ours:.2e} from float64 (reference: {e_ref:.2e})"
        elif name.startswith("kpf_encoder."):
            n_enc += 1
            enc_loose += err > 1e-4
            assert nerr <= 5e-4, f"{tag}/{which} {name}: norm deviates by {nerr:.2e}"
            assert rms <= 2e-2, f"{tag}/{which} {name}: RMS deviation {rms:.2e}"
        else:
            assert nerr <= 1e-4, f"{tag}/{which} {name}: norm deviates by {nerr:.2e}"
            assert err <= 1e-4, f"{tag}/{which} {name}: entries deviate by {err:.2e}"
        report.append((err, name))
    report.sort(reverse=True)
    print(f"{tag}/{which}: {n_checked} tensors; encoder tensors above 1e-4 entrywise: {enc_loose}/{n_enc}; worst: " +
          ", ".join(f"{n} {e:.1e}" for e, n in report[:3]))
    assert n_checked >= 100
    # 3dmatch: the neighbour matrices of levels 1 and 2 are cut at limit = 40 and a tie at the cut
    # keeps a different (equally distant) neighbour than the reference's kd-tree order
    # (tests/test_gpu_preprocess.py) -- one more discrete difference on top of the branch flips,
    # so the entry-wise count is not asserted there (norm and RMS bounds above still are)
    if tag != "3dmatch":
        assert enc_loose <= 0.5 * n_enc, f"{enc_loose} of {n_enc} encoder tensors deviate entrywise"


def _frozen_encoder_gradient_check(device, tag, clouds, bound, conditioned=()):
    cfg = get_config(tag)
    model = RegTR(cfg)
    synthetic.fill_parameters(model, seed=0)
    model = model.to(device).train()
    pts = [T(c).to(device) for c in clouds]
    meta = model.preprocessor(pts)
    enc = model.kpf_encoder
    # every LeakyReLU site of the encoder, in the oracle's call order: SimpleBlock output; per
    # ResNet block unary1 (when present), batch_norm_conv, block output
    acts, hooks = [], []
    for blk in enc.encoder_blocks:
        if hasattr(blk, 'unary1') and not isinstance(blk.unary1, torch.nn.Identity):
            hooks.append(blk.unary1.register_forward_hook(lambda m, a, o: acts.append(o.detach())))
        if hasattr(blk, 'batch_norm_conv'):
            hooks.append(blk.batch_norm_conv.register_forward_hook(lambda m, a, o: acts.append(o.detach())))
        hooks.append(blk.register_forward_hook(lambda m, a, o: acts.append(o.detach())))
    block_in = []
    for blk in enc.encoder_blocks:
        hooks.append(blk.register_forward_pre_hook(lambda m, a: block_in.append(a[0].detach())))
    x0 = torch.ones((meta['points'][0].shape[0], 1), device=device)
    f, _ = enc(x0, meta)
    for h in hooks:
        h.remove()
    G = synthetic.rand(tuple(f.shape), 77, -1.0, 1.0).to(device)
    enc.zero_grad(set_to_none=True)
    (f * G).sum().backward()
    # ---- the same objective through the float64 oracle with our decisions ------------------------
    meta64 = {'points': [p.double().cpu() for p in meta['points']],
              'stack_lengths': [l.cpu() for l in meta['stack_lengths']],
              'neighbors': [n.long().cpu() for n in meta['neighbors']],
              'pools': [n.long().cpu() for n in meta['pools']]}
    pool_args = []
    for i, name in enumerate(cfg.architecture):
        if 'strided' in name and 'resnetb' in name:
            lvl = sum(1 for n in cfg.architecture[:i] if 'strided' in n or 'pool' in n)
            xin = block_in[i].cpu()
            x_ext = torch.cat((xin, torch.zeros_like(xin[:1])), 0)
            pool_args.append(x_ext[meta64['pools'][lvl]].max(1)[1])       # first maximum, like spr_maxpool_bwd
    sd = {k: v.detach().double().cpu().requires_grad_(v.requires_grad and 'kernel_points' not in k)
          for k, v in model.state_dict(keep_vars=True).items() if k.startswith('kpf_encoder.')}
    frozen = O.FrozenDecisions([a.cpu() > 0 for a in acts], pool_args)
    f64, feats64 = O.encoder(cfg, sd, meta64, frozen=frozen)
    assert frozen.i_mask == len(acts) and frozen.i_pool == len(pool_args)
    assert float((f.detach().double().cpu() - f64.detach()).abs().max()) <= 2e-5 * float(f64.detach().abs().max())
    g_first = []
    feats64[0].register_hook(lambda g: g_first.append(g.detach().clone()))
    (f64 * G.double().cpu()).sum().backward(retain_graph=bool(conditioned))
    ref_grads = {n: sd[n].grad.clone() for n in sd if sd[n].grad is not None}
    # Conditioning floor of the tensors named in `conditioned` (parameters of the first block): how far
    # does the float64 result move when the gradient arriving at the first block's output carries a
    # RELATIVE error of 2e-6 -- what every other tensor of this very test shows our float32 backward
    # chain to have?  (The gradient is linear in it: one more backward with the perturbation alone.)
    floor = {}
    if conditioned:
        for n in sd:
            if sd[n].grad is not None:
                sd[n].grad = None
        noise = torch.randn(g_first[0].shape, dtype=torch.float64, generator=torch.Generator().manual_seed(5))
        feats64[0].backward(gradient=2e-6 * g_first[0].abs() * noise)
        for n in conditioned:
            floor[n] = sd[n].grad.abs().max()
    worst, n_checked, excused = [], 0, []
    for name, p in model.named_parameters():
        if not name.startswith('kpf_encoder.') or not p.requires_grad:
            continue
        ref = ref_grads[name]
        assert p.grad is not None, name
        err = float((p.grad.double().cpu() - ref).abs().max())
        scale = max(float(ref.abs().max()), float(ref.norm()) / math.sqrt(ref.numel()), 1e-30)
        n_checked += 1
        if name in floor and err / scale > bound:
            # an ill-conditioned sum: held to 4x what a 2e-6 relative error of its INPUT does to it
            assert err <= 4.0 * float(floor[name]), f"{tag}: {name} deviates by {err / scale:.2e}, conditioning floor {float(floor[name]) / scale:.2e}"
            excused.append((name, err / scale, float(floor[name]) / scale))
            continue
        worst.append((err / scale, name))
    worst.sort(reverse=True)
    for name, e, fl in excused:
        print(f"{tag}: {name} deviates {e:.1e} of its scale; a 2e-6 relative perturbation of its input alone moves it by {fl:.1e}")
    print(f"{tag}: {n_checked} encoder tensors, worst entrywise deviations: " + ", ".join(f"{n} {e:.1e}" for e, n in worst[:3]))
    assert n_checked >= 15
    assert worst[0][0] <= bound, f"{tag}: {worst[0][1]} deviates entrywise by {worst[0][0]:.2e}"


@pytest.mark.parametrize("tag", ["3dmatch", "kitti", "modelnet"])
def test_encoder_gradients_entrywise_with_frozen_decisions(device, tag):
    """The KPConv encoder's parameter gradients ENTRY BY ENTRY at 1e-4, on all three configs, with
    the discrete decisions frozen.

    The reference-gradient test above can only bound encoder tensors by norm + RMS because a forward
    difference of 1e-6 flips LeakyReLU branches / max-pool winners.  Here those decisions are taken
    from OUR forward (sign of every activation the HIP path produced, arg-max neighbour of every
    max-pool on our activations, our neighbour matrices) and replayed inside the float64 oracle
    (oracle.torch_oracle.FrozenDecisions): both sides then differentiate the same smooth function,
    an indexing or scatter error in spr_kpconv_bwd_dx / spr_instnorm_bwd / spr_maxpool_bwd would show
    as an O(1) entry error, and every encoder tensor is held to 2e-5 of its scale entrywise (measured:
    2.2e-6 / 3.5e-6 / 2.5e-6 on the three configs)."""
    pairs, sizes = pairs_for(tag, 2)
    clouds = [p[0][:n] for p, (n, m) in zip(pairs, sizes)] + [p[1][:m] for p, (n, m) in zip(pairs, sizes)]
    _frozen_encoder_gradient_check(device, tag, clouds, 2e-5)


def test_encoder_gradients_at_baseline_size(device):
    """The same frozen-decision comparison at BASELINE configs[1] size: one 16 384-point pair of the bench
    generator (32 768 points at level 0, ~13 k / 3.9 k at levels 1 / 2) through the whole KPConv encoder --
    the backward kernels at the shapes the bench runs (large bgemm tiles, split-K weight gradients over
    hundreds of slabs, the KPConv dx scatter with ~40-wide rows) against float64 autograd of the oracle
    with our decisions.  Every encoder tensor entrywise within 5e-5 of its scale (measured: 2e-6).
    (At this size the test found an activation that is EXACTLY zero -- x equal to the rounded column mean,
    once in ~10^7 elements -- where spr_instnorm_bwd took the identity branch and torch's leaky_relu
    backward the slope branch: one element of one gradient off by a factor 10, 2.5e-3 of the first
    KPConv's weight gradient.  Fixed in norm_pool.hip; the first layer's weight gradient, a nearly
    cancelling sum over all points, is also accumulated in float64 now, spr_tn_product_f64.)"""
    src, tgt, _ = synthetic.make_pair(16384, seed=0)
    _frozen_encoder_gradient_check(device, "3dmatch", [src, tgt], 5e-5)


def test_two_training_steps_match_the_reference_loop(device):
    """training.Trainer on the HIP model vs the reference's own loop (trainer.py:107-124 with the
    optimiser of generic_reg_model.py:46-76) for two steps on one batch: losses, the clipped
    gradient norm implied by the update, and the parameter UPDATES at the pinned entries
    (golden: tests/golden/train_3dmatch_b2.npz, oracle/gen_golden.py gen_train).  Also the
    validation metrics of the golden poses (generic_reg_model.py:294-321)."""
    from superpoints_registration_amd.training import Trainer, compute_metrics
    g = load_golden("train_3dmatch_b2.npz")
    tag, B = "3dmatch", int(g["B"])
    cfg = get_config(tag)
    pairs, sizes = pairs_for(tag, B)
    pose, src_ov, tgt_ov = loss_inputs(tag, B)
    model = RegTR(cfg)
    synthetic.fill_parameters(model, seed=0)
    model = model.to(device)
    batch = {"src_xyz": [T(p[0][:n]).to(device) for p, (n, m) in zip(pairs, sizes)],
             "tgt_xyz": [T(p[1][:m]).to(device) for p, (n, m) in zip(pairs, sizes)],
             "pose": T(pose).to(device),
             "src_overlap": [T(o).to(device) for o in src_ov], "tgt_overlap": [T(o).to(device) for o in tgt_ov]}
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    tr = Trainer(cfg).setup(model)
    for st in range(int(g["steps"])):
        losses = tr.train_step(model, dict(batch))
        ref = float(g[f"step{st}_total"])
        assert abs(float(losses["total"]) - ref) <= 2e-4 * abs(ref), (st, float(losses["total"]), ref)
    # AdamW's first steps move every entry by ~lr * sign-like terms: compare updates entry-wise,
    # relative to the update scale lr = 1e-4 (a wrong step order -- no clipping, or clipping after
    # the step -- changes the second step's m / sqrt(v) ratios by far more than this tolerance)
    worst = 0.0
    for n, p in model.named_parameters():
        if not p.requires_grad:
            continue
        d = (p.detach() - before[n]).double().reshape(-1).cpu().numpy()
        got = d[grad_sample_indices(n, d.size)]
        refd = g[f"delta|{n}"].astype(np.float64)
        dev = np.abs(got - refd)
        # entries whose gradient is ~0 change sign freely under AdamW (update = lr * m / (sqrt(v) + eps)):
        # require 85 % of the (64) pinned entries within 2 % of the two-step size and the pinned
        # update vectors to be collinear (cosine >= 0.995)
        frac = float((dev <= 0.02 * 2e-4).mean())
        if not np.any(refd):                      # no gradient in the reference (unused loss head): no update
            assert not np.any(got), n
            continue
        cos = float(np.dot(got, refd) / max(np.linalg.norm(got) * np.linalg.norm(refd), 1e-30))
        worst = max(worst, 1 - frac)
        assert d.size <= 4 or (frac >= 0.85 and cos >= 0.995), f"{n}: {frac:.3f} of the pinned updates agree, cosine {cos:.4f}"
    print(f"two training steps: worst tensor has {worst * 100:.2f} % of pinned updates off")
    # validation metrics on the golden forward (eval mode, untouched parameters)
    model2 = RegTR(cfg)
    synthetic.fill_parameters(model2, seed=0)
    model2 = model2.to(device).eval()
    out = model2({"src_xyz": batch["src_xyz"], "tgt_xyz": batch["tgt_xyz"]})
    m = compute_metrics(out, {"pose": batch["pose"]})
    assert np.allclose(m["rot_err_deg"].cpu().numpy(), g["rot_err_deg"], atol=2e-2)     # degrees
    assert np.allclose(m["trans_err"].cpu().numpy(), g["trans_err"], atol=1e-4)
