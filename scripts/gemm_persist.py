"""Persistent GEMM (stores overlapped with the next tile's K loop) vs the plain tiled kernel:
accuracy against float64 and time per call.  SPR_GEMM_PERSIST=0 selects the plain kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superpoints_registration_amd import ops
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
for (m, k, n, act) in ((61745, 256, 1024, 1), (61745, 256, 256, 0), (61745, 256, 768, 0), (5000, 256, 300, 0), (2048, 256, 256, 1)):
    x = torch.randn(m, k, generator=g).to(dev); w = (torch.randn(n, k, generator=g) * 0.05).to(dev); b = (torch.randn(n, generator=g) * 0.1).to(dev)
    y = ops.linear(x, w, b, act=act)
    ref = x.double() @ w.double().t() + b.double()
    if act == 1: ref = ref.clamp_min(0)
    err = float((y.double() - ref).abs().max() / ref.abs().max())
    rng = getattr(y, '_spr_range', None)
    rmax = float(rng[0][:rng[1]].max()) if rng is not None else float('nan')
    for _ in range(3): ops.linear(x, w, b, act=act)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.linear(x, w, b, act=act)
    e1.record(); torch.cuda.synchronize()
    print('persist', os.environ.get('SPR_GEMM_PERSIST', '1'), (m, k, n, act), 'rel err %.2e' % err, 'published max %.4f true %.4f' % (rmax, float(y.abs().max())),
          'us/call %.1f' % (e0.elapsed_time(e1) * 50))
