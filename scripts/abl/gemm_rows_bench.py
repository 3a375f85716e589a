"""Row-streaming GEMM (k_gemm_rows) vs the tiled kernels on the encoder's skinny products: time, GB/s, max error."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superpoints_registration_amd import ops, synthetic
dev = torch.device('cuda')
def t(fn, reps=10):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for m, k, n in ((524288, 64, 32), (524288, 32, 128), (524288, 64, 128), (524288, 128, 32), (211914, 32, 128), (211914, 128, 64), (5000, 64, 32), (211913, 64, 64)):
    x = (torch.rand((m, k), device=dev) - 0.4) * 3
    w = (torch.rand((n, k), device=dev) - 0.5) * 0.3
    y = ops.linear_raw(x, w)
    ref = (x.double() @ w.double().t())
    err = float((y.double() - ref).abs().max() / ref.abs().max())
    us = t(lambda: ops.linear_raw(x, w))
    gb = (m * k + m * n) * 4 / 1e9
    print('m=%d %d->%d: %.1f us  %.2f TB/s  err %.2e  (%s)' % (m, k, n, us, gb / us * 1e3 / 1e3 * 1e3 / 1e3 if False else gb / (us * 1e-6) / 1e3, err, os.environ.get('SPR_GEMM_ROWS', '1')), flush=True)
