"""Time line of the ring KPConv kernel (experiment build with -DSPR_KP_RING_PROF, loaded through
SPR_HIP_LIB): workgroup 0's eight waves stamp the shader clock at phase boundaries and item starts
for their first tiles; prints per-tile phase durations per wave and item-duration statistics."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
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
N = 160
names = {1: 'P1 start', 2: 'item', 3: 'items done', 4: 'primed', 5: 'past B1', 6: 'P2 done', 7: 'past B2', 8: 'epilogue done'}
for lvl, c in ((1, 64), (0, 32)):
    nb = meta['_i32'][('neighbors', lvl)]; pts = meta['points'][lvl]
    x = torch.rand((pts.shape[0], c), device=dev) - 0.3
    W = (torch.rand((15, c, c), device=dev) - 0.5) * 0.2
    ext = cfg.first_subsampling_dl * cfg.KP_extent * (2 ** lvl)
    f = lambda: ops.kpconv_raw(pts, pts, nb, x, W, KP * (2 ** lvl), ext, rows_sorted=True)
    f(); f(); torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (8 * N))()
    assert L.spr_debug_kp_trace(buf, 8 * N) == 0
    tr = np.array(list(buf), dtype=np.uint64).reshape(8, N)
    print('== level %d %d->%d' % (lvl, c, c))
    for w in range(8):
        ids = (tr[w] & np.uint64(15)).astype(int); ts = (tr[w] >> np.uint64(4)).astype(np.int64)
        k = int((ids > 0).sum()); ids, ts = ids[:k], ts[:k]
        # split into tiles at id 1
        starts = [i for i in range(k) if ids[i] == 1]
        line = []
        for a, b in zip(starts, starts[1:] + [k]):
            seg_ids, seg_ts = ids[a:b], ts[a:b]
            if 8 not in seg_ids: break
            t = {i: seg_ts[list(seg_ids).index(i)] for i in (1, 3, 4, 5, 6, 7, 8) if i in seg_ids}
            items = seg_ts[seg_ids == 2]
            n_it = len(items)
            per = (t[3] - t[1]) / max(n_it, 1)
            line.append('[%d it %5d/it | pr %4d b1 %5d p2 %4d b2 %4d ep %4d]' % (
                n_it, per, t[4] - t[3], t[5] - t[4], t[6] - t[5], t[7] - t[6], t[8] - t[7]))
        print(' wave %d: %s' % (w, ' '.join(line[:5])))
    # item-to-item deltas of wave 0
    ids = (tr[0] & np.uint64(15)).astype(int); ts = (tr[0] >> np.uint64(4)).astype(np.int64)
    it = ts[ids == 2]
    if len(it) > 3:
        print(' wave 0 item start deltas:', np.diff(it)[:40].tolist())
