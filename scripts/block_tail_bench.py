"""Fused ResNet-block tail (spr_block_tail) vs the separate operators it replaces, at the bench's shapes:
time per call and the HBM bytes each form moves (algorithmic: inputs + output once for the fused form)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superpoints_registration_amd import ops
dev = torch.device('cuda')
def t(fn, reps=10):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
# (rows, clouds, ka, kb, n_out): the seven bottleneck blocks of the 3dmatch encoder on 16 pairs x 16384 points
CASES = [(524288, 32, 32, 64, 128), (211914, 32, 32, 0, 128), (211914, 32, 64, 128, 256), (211914, 32, 64, 0, 256),
         (61745, 32, 64, 0, 256), (61745, 32, 128, 256, 512), (61745, 32, 128, 0, 512)]
for n, nb, ka, kb, no in CASES:
    lens = [n // nb + (1 if i < n % nb else 0) for i in range(nb)]
    cu = ops.lengths_to_cu(lens, dev)
    xa = torch.randn((n, ka), device=dev); wa = torch.randn((no, ka), device=dev) * 0.1
    xb = wb = ad = None
    if kb: xb = torch.randn((n, kb), device=dev); wb = torch.randn((no, kb), device=dev) * 0.1
    else: ad = torch.randn((n, no), device=dev)
    def fused(): return ops.block_tail(xa, wa, cu, xb=xb, wb=wb, add=ad)
    def sep():
        sc = ad if xb is None else ops.instnorm_raw(ops.linear_raw(xb, wb), cu, max_len=max(lens))
        return ops.instnorm_raw(ops.linear_raw(xa, wa), cu, add=sc, slope=0.1, max_len=max(lens))
    err = float((fused() - sep()).abs().max())
    tf, ts = t(fused), t(sep)
    gb = n * 4 * (2 * (ka + kb) + no + (no if ad is not None else 0)) / 1e9
    print('n=%d %d+%d->%d: fused %.1f us (%.2f TB/s of %.2f GB)  separate %.1f us  x%.2f  maxdiff %.1e' % (
        n, ka, kb, no, tf, gb / tf * 1e3, gb, ts, ts / tf, err), flush=True)
