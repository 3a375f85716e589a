"""KPConv timing ablations through the public op (input-side): which resource bounds the kernel?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superpoints_registration_amd import get_config, ops, synthetic
from superpoints_registration_amd.regtr import RegTR

dev = torch.device('cuda:0')
B = 16
cfg = get_config('3dmatch')
model = RegTR(cfg); synthetic.fill_parameters(model, 0); model = model.to(dev).eval()
pairs = [synthetic.make_pair(16384, seed=i) for i in range(B)]
meta = model.preprocessor([torch.from_numpy(p[0]).to(dev) for p in pairs] + [torch.from_numpy(p[1]).to(dev) for p in pairs])
pts = meta['points'][1]; nb = meta['_i32'][('neighbors', 1)]
n = pts.shape[0]
print('L1 points', n, 'nbr', tuple(nb.shape), 'valid frac', float((nb < n).float().mean()))
blk = model.kpf_encoder.encoder_blocks[3].KPConv   # 64 -> 64
x = torch.rand((n, 64), device=dev) - 0.3
W, KP, ext = blk.weights.detach(), blk.kernel_points.detach(), blk.KP_extent

def t(fn, reps=10):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

K = nb.shape[1]
self_idx = torch.arange(n, device=dev, dtype=torch.int32)[:, None].expand(-1, K).contiguous()
self_masked = torch.where(nb < n, self_idx, torch.full_like(nb, n))
local = torch.where(nb < n, (self_idx + torch.arange(K, device=dev, dtype=torch.int32)[None, :]).clamp(max=n - 1), torch.full_like(nb, n))
print('normal                       %.0f us' % t(lambda: ops.kpconv(pts, pts, nb, x, W, KP, ext, rows_sorted=True)))
print('gather self (same validity)  %.0f us' % t(lambda: ops.kpconv(pts, pts, self_masked, x, W, KP, ext, rows_sorted=True)))
print('gather n+k (streaming rows)  %.0f us' % t(lambda: ops.kpconv(pts, pts, local, x, W, KP, ext, rows_sorted=True)))
print('first 16 columns only        %.0f us' % t(lambda: ops.kpconv(pts, pts, nb[:, :16], x, W, KP, ext, rows_sorted=True)))
one = torch.full_like(nb, n); one[:, 0] = self_idx[:, 0]
print('1 valid neighbour per row    %.0f us' % t(lambda: ops.kpconv(pts, pts, one, x, W, KP, ext, rows_sorted=True)))
for cout in (64, 128, 256):
    W2 = (torch.rand((15, 64, cout), device=dev) - 0.5) * 0.2
    print('cout=%d normal               %.0f us' % (cout, t(lambda: ops.kpconv(pts, pts, nb, x, W2, KP, ext, rows_sorted=True))))
    print('cout=%d 1 valid nbr          %.0f us' % (cout, t(lambda: ops.kpconv(pts, pts, one, x, W2, KP, ext, rows_sorted=True))))
