"""Ring kernel (impl 0) vs the round-2 streamed kernel (impl 2) on the bench pyramid: time per launch,
algorithmic bytes, roofline fraction, and the largest difference between the two outputs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superpoints_registration_amd import get_config, ops, synthetic
from superpoints_registration_amd.regtr import RegTR

dev = torch.device('cuda:0')
B = int(os.environ.get('PAIRS', 16))
cfg = get_config('3dmatch')
model = RegTR(cfg); synthetic.fill_parameters(model, 0); model = model.to(dev).eval()
pairs = [synthetic.make_pair(16384, seed=i) for i in range(B)]
meta = model.preprocessor([torch.from_numpy(p[0]).to(dev) for p in pairs] + [torch.from_numpy(p[1]).to(dev) for p in pairs])


def t(fn, reps=10):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


KP = model.kpf_encoder.encoder_blocks[1].KPConv.kernel_points.detach()
LEVELS = [int(v) for v in os.environ.get('LEVELS', '0,1,2').split(',')]
KINDS = os.environ.get('KINDS', 'neighbors,pools').split(',')
for lvl, c in ((0, 32), (1, 64), (2, 128)):
    if lvl not in LEVELS:
        continue
    for kind in KINDS:
        if (kind, lvl) not in meta['_i32']:
            continue
        nb = meta['_i32'][(kind, lvl)]
        s_pts = meta['points'][lvl]
        q_pts = s_pts if kind == 'neighbors' else meta['points'][lvl + 1]
        ns, nq = s_pts.shape[0], q_pts.shape[0]
        x = torch.rand((ns, c), device=dev) - 0.3
        W = (torch.rand((15, c, c), device=dev) - 0.5) * 0.2
        ext = cfg.first_subsampling_dl * cfg.KP_extent * (2 ** lvl)
        kv = int((nb < ns).sum())
        alg = kv * (4 + 12 + 4 * c) + nq * (12 + 4 * c) + 60 * c * c
        res = {}
        for impl in (0, 2):
            f = lambda: ops.kpconv_raw(q_pts, s_pts, nb, x, W, KP * (2 ** lvl), ext, rows_sorted=True, impl=impl)
            y = f(); torch.cuda.synchronize()
            res[impl] = (t(f), y)
        if os.environ.get('ORDER'):
            # spatial tile walk: queries sorted by the linear index of their cell (cell = conv radius), cloud by cloud
            lens = meta['stack_lengths'][lvl if kind == 'neighbors' else lvl + 1].to(dev)
            cloud = torch.repeat_interleave(torch.arange(len(lens), device=dev), lens.long())
            r = cfg.first_subsampling_dl * cfg.conv_radius * (2 ** lvl)
            cc = (q_pts / r).floor().long(); cc -= cc.min(0)[0]
            if os.environ['ORDER'] == 'morton':
                code = torch.zeros(nq, dtype=torch.long, device=dev)
                for bb in range(10):
                    for a in range(3):
                        code |= ((cc[:, a] >> bb) & 1) << (3 * bb + a)
            else:
                code = (cc[:, 2] * 4096 + cc[:, 1]) * 4096 + cc[:, 0]
            order = torch.argsort(cloud * (1 << 40) + code).to(torch.int32)
            fo = lambda: ops.kpconv_raw(q_pts, s_pts, nb, x, W, KP * (2 ** lvl), ext, rows_sorted=True, order=order)
            yo = fo(); torch.cuda.synchronize()
            print('   with %s tile order: %.1f us (bitwise equal: %s)' % (os.environ['ORDER'], t(fo), bool(torch.equal(yo, res[0][1]))), flush=True)
        d = float((res[0][1] - res[2][1]).abs().max()); sc = float(res[2][1].abs().max())
        print('L%d %-9s %3d->%3d nq %7d kv %8d alg %.3f GB | ring %7.1f us (%.3f of 8 TB/s) | r2 %7.1f us (%.3f) | max diff %.2e of %.2e' % (
            lvl, kind, c, c, nq, kv, alg / 1e9, res[0][0], alg / res[0][0] / 1e6 / 8.0, res[2][0], alg / res[2][0] / 1e6 / 8.0, d, sc), flush=True)
