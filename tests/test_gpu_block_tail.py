"""GPU: the fused tail of the ResNet bottleneck block (spr_block_tail; kpconv_blocks.py:733-741) against
a float64 evaluation of the reference's expression and against the separate operators it replaces
(spr_linear_r + spr_instnorm_r), on ragged batches with clouds shorter than, equal to and not a
multiple of the 64-row statistics tile.  Tolerances are written next to each check."""
import numpy as np
import pytest
import torch

from superpoints_registration_amd import ops

pytestmark = pytest.mark.gpu

# (ka, kb, n_out): every shape with a kernel (csrc/block_tail.hip tail_shape)
SHAPES = [(32, 64, 128), (32, 0, 128), (64, 128, 256), (64, 0, 256), (128, 256, 512), (128, 0, 512)]
LENS = [1, 63, 64, 65, 700, 129, 2048, 17, 3001]


def _inputs(ka, kb, n_out, lens, device, seed=0, add=True, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    n = sum(lens)
    xa = ((torch.rand((n, ka), generator=g) - 0.3) * 3 * scale).to(device)
    wa = ((torch.rand((n_out, ka), generator=g) - 0.5) * 0.4).to(device)
    xb = wb = ad = None
    if kb > 0:
        xb = ((torch.randn((n, kb), generator=g) * 0.7 + 0.2) * scale).to(device)
        wb = ((torch.rand((n_out, kb), generator=g) - 0.5) * 0.2).to(device)
    elif add:
        ad = torch.randn((n, n_out), generator=g).to(device)
    cu = ops.lengths_to_cu(lens, device)
    return xa, wa, xb, wb, ad, cu


def _ref64(xa, wa, xb, wb, ad, lens, eps=1e-5, slope=0.1):
    """kpconv_blocks.py:556-561 + :497-525 + :741 in float64."""
    def inorm(y):
        out = torch.empty_like(y)
        o = 0
        for l in lens:
            s = y[o:o + l]
            m = s.mean(0, keepdim=True)
            v = ((s - m) ** 2).mean(0, keepdim=True)
            out[o:o + l] = (s - m) / torch.sqrt(v + eps)
            o += l
        return out
    y = inorm(xa.double() @ wa.double().t())
    if xb is not None:
        y = y + inorm(xb.double() @ wb.double().t())
    elif ad is not None:
        y = y + ad.double()
    return torch.where(y >= 0, y, y * slope)


def _unfused(xa, wa, xb, wb, ad, cu, lens):
    sc = ad
    if xb is not None:
        sc = ops.instnorm_raw(ops.linear_raw(xb, wb), cu, max_len=max(lens))
    return ops.instnorm_raw(ops.linear_raw(xa, wa), cu, add=sc, slope=0.1, max_len=max(lens))


@pytest.mark.parametrize("shape", SHAPES)
def test_block_tail_vs_float64_and_separate_operators(device, shape):
    ka, kb, n_out = shape
    assert ops.block_tail_tile_rows(ka, kb, n_out) > 0
    xa, wa, xb, wb, ad, cu = _inputs(ka, kb, n_out, LENS, device)
    out = ops.block_tail(xa, wa, cu, xb=xb, wb=wb, add=ad)
    ref = _ref64(xa.cpu(), wa.cpu(), None if xb is None else xb.cpu(), None if wb is None else wb.cpu(),
                 None if ad is None else ad.cpu(), LENS)
    # rows of clouds with >= 17 points: 1e-5 of the output scale (normalised values are O(1); the products
    # carry 22-bit operands, the statistics are float64)
    long_rows = torch.cat([torch.full((l,), l >= 17) for l in LENS])
    err = (out.cpu().double() - ref).abs()
    assert float(err[long_rows].max()) <= 1e-5 * float(ref.abs().max()), float(err[long_rows].max())
    # the 1-point cloud: variance 0 -> (y - y) * rstd = 0 exactly, like nn.InstanceNorm1d on one sample
    one = out[:1].cpu().double()
    assert float((one - ref[:1]).abs().max()) <= 1e-5
    # against the operators it replaces: same arithmetic up to the order of the float64 sums and the
    # MFMA shape of the products -> 2e-6 of the output scale
    sep = _unfused(xa, wa, xb, wb, ad, cu, LENS)
    assert float((out - sep).abs()[long_rows.to(device)].max()) <= 2e-6 * float(sep.abs().max())
    # published range: an upper bound of max |out| within a factor of two
    parts, n = ops._get_range(out)
    assert parts is not None
    pub = float(parts[:n].max())
    assert pub == float(out.abs().max())


def test_block_tail_without_shortcut_tensor(device):
    xa, wa, _, _, _, cu = _inputs(32, 0, 128, LENS, device, add=False)
    out = ops.block_tail(xa, wa, cu)
    ref = _ref64(xa.cpu(), wa.cpu(), None, None, None, LENS)
    assert float((out.cpu().double() - ref).abs()[64:].max()) <= 1e-5 * float(ref.abs().max())


@pytest.mark.parametrize("shape", [(32, 64, 128), (64, 0, 256), (128, 256, 512)])
def test_block_tail_is_batch_invariant_bitwise(device, shape):
    """Tiles start at each cloud's first row: copies of a cloud give bitwise the same rows wherever they
    sit in the batch and however many there are (the per-tensor operand scales are equal by construction)."""
    ka, kb, n_out = shape
    l = 777
    xa, wa, xb, wb, ad, _ = _inputs(ka, kb, n_out, [l], device, seed=5)
    rep = lambda t, k: None if t is None else t.repeat(k, 1)
    outs = []
    for k in (1, 2, 5):
        cu = ops.lengths_to_cu([l] * k, device)
        outs.append(ops.block_tail(rep(xa, k), wa, cu, xb=rep(xb, k), wb=wb, add=rep(ad, k)))
    for k, o in zip((1, 2, 5), outs):
        for c in range(k):
            assert torch.equal(o[c * l:(c + 1) * l], outs[0])
    # and run to run
    cu = ops.lengths_to_cu([l] * 5, device)
    again = ops.block_tail(rep(xa, 5), wa, cu, xb=rep(xb, 5), wb=wb, add=rep(ad, 5))
    assert torch.equal(again, outs[2])


@pytest.mark.parametrize("scale", [1e-5, 1e3])
def test_block_tail_magnitudes(device, scale):
    """Range-scaled operands: inputs of magnitude 1e-5 and 1e3 give the same normalised output accuracy."""
    lens = [300, 1000]
    xa, wa, xb, wb, ad, cu = _inputs(64, 128, 256, lens, device, seed=2, scale=scale)
    out = ops.block_tail(xa, wa, cu, xb=xb, wb=wb, eps=0.0 if scale < 1 else 1e-5)
    ref = _ref64(xa.cpu(), wa.cpu(), xb.cpu(), wb.cpu(), None, lens, eps=0.0 if scale < 1 else 1e-5)
    assert float((out.cpu().double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())


def test_block_tail_many_small_clouds(device):
    """ModelNet-like: 512 clouds of 90..200 rows (several partially filled tiles per cloud)."""
    rng = np.random.default_rng(1)
    lens = [int(v) for v in rng.integers(90, 200, 512)]
    xa, wa, xb, wb, ad, cu = _inputs(128, 0, 512, lens, device, seed=3)
    out = ops.block_tail(xa, wa, cu, add=ad)
    sep = _unfused(xa, wa, None, None, ad, cu, lens)
    assert float((out - sep).abs().max()) <= 2e-6 * float(sep.abs().max())


def test_block_tail_rejects_unsupported_shape(device):
    assert ops.block_tail_tile_rows(48, 0, 128) == 0
    xa = torch.zeros((10, 48), device=device)
    wa = torch.zeros((128, 48), device=device)
    with pytest.raises(RuntimeError):
        ops.block_tail(xa, wa, ops.lengths_to_cu([10], device))


@pytest.mark.parametrize("shape", SHAPES)
def test_block_tail_normalises_its_input_on_load(device, shape):
    """Round 5 (spr_block_tail_n): xa handed over RAW with the statistics of the per-cloud InstanceNorm +
    LeakyReLU(0.1) that precedes the tail in a bottleneck block (kpconv_blocks.py:717-719 of the reference).  Against
    the float64 expression lrelu(IN(lrelu(IN(xa)) wa^T) + ...) at 5e-6 of the output scale, and against the two
    separate operators (instnorm, then the plain tail) at 2e-6: the only difference between the two GPU routes is
    the operand scale of the staged tile (the static bound sqrt(longest cloud) instead of the measured maximum)."""
    ka, kb, n_out = shape
    lens = LENS
    xa, wa, xb, wb, ad, cu = _inputs(ka, kb, n_out, lens, device, seed=3)
    mean, rstd = ops.instnorm_stats(xa, cu, max_len=max(lens))
    got = ops.block_tail(xa, wa, cu, xb=xb, wb=wb, add=ad, xa_stats=(mean, rstd), xa_slope=0.1, xa_max_len=max(lens))
    xn = ops.instnorm_raw(xa, cu, slope=0.1, max_len=max(lens))
    two = ops.block_tail(xn, wa, cu, xb=xb, wb=wb, add=ad)

    def inorm(y):
        out = torch.empty_like(y)
        o = 0
        for l in lens:
            sl = y[o:o + l]
            m = sl.mean(0, keepdim=True)
            v = ((sl - m) ** 2).mean(0, keepdim=True)
            out[o:o + l] = (sl - m) / torch.sqrt(v + 1e-5)
            o += l
        return out
    x64 = inorm(xa.double().cpu())
    x64 = torch.where(x64 >= 0, x64, x64 * 0.1)
    ref = _ref64(x64, wa.cpu(), None if xb is None else xb.cpu(), None if wb is None else wb.cpu(),
                 None if ad is None else ad.cpu(), lens)
    scale = float(ref.abs().max())
    assert torch.isfinite(got).all()
    assert float((got.double().cpu() - ref).abs().max()) <= 5e-6 * scale
    assert float((got - two).abs().max()) <= 2e-6 * scale
    # the statistics entry point is the norm operator's own first two passes
    m64 = torch.stack([xa.double().cpu()[o:o + l].mean(0) for o, l in zip(np.concatenate([[0], np.cumsum(lens)[:-1]]), lens)])
    assert float((mean.double().cpu() - m64).abs().max()) <= 1e-6 * float(xa.abs().max())
