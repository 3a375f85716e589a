"""Phase breakdown of the ring KPConv kernel (experiment build with -DSPR_KP_RING_PROF, loaded through
SPR_HIP_LIB): shader-clock totals per phase, summed over waves, as a share of the waves' lifetime."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superpoints_registration_amd import get_config, ops, synthetic, _lib
from superpoints_registration_amd.regtr import RegTR

dev = torch.device('cuda:0')
cfg = get_config('3dmatch')
model = RegTR(cfg); synthetic.fill_parameters(model, 0); model = model.to(dev).eval()
pairs = [synthetic.make_pair(16384, seed=i) for i in range(16)]
meta = model.preprocessor([torch.from_numpy(p[0]).to(dev) for p in pairs] + [torch.from_numpy(p[1]).to(dev) for p in pairs])
L = ctypes.CDLL(_lib.LIB_PATH)
KP = model.kpf_encoder.encoder_blocks[1].KPConv.kernel_points.detach()
names = ['phase 1 (total)', '  waiting for the ring', '  issue', 'stage + prime', 'barrier 1', 'phase 2', 'barrier 2',
         'epilogue']
for lvl, c in ((1, 64), (0, 32)):
    nb = meta['_i32'][('neighbors', lvl)]; pts = meta['points'][lvl]
    x = torch.rand((pts.shape[0], c), device=dev) - 0.3
    W = (torch.rand((15, c, c), device=dev) - 0.5) * 0.2
    ext = cfg.first_subsampling_dl * cfg.KP_extent * (2 ** lvl)
    f = lambda: ops.kpconv_raw(pts, pts, nb, x, W, KP * (2 ** lvl), ext, rows_sorted=True)
    f(); torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    L.spr_debug_kp_prof(buf, 1)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); f(); b.record(); torch.cuda.synchronize()
    L.spr_debug_kp_prof(buf, 1)
    v = list(buf)
    waves = 256 * 8
    tot = sum(v[k] for k in (0, 3, 4, 5, 6, 7))
    print('level %d %d->%d: op %.1f us (with stamps); per wave: %.0f cycles accounted, %d items, %d tiles' % (
        lvl, c, c, a.elapsed_time(b) * 1e3, tot / waves, v[8] / waves, v[9] / waves))
    for k, n in enumerate(names):
        print('   %-26s %6.1f %%   %8.0f cycles per tile' % (n, 100.0 * v[k] / tot, v[k] / max(v[9], 1)))
