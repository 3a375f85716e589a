"""Per-launch KPConv kernel times inside the full model forward (HIP events inside spr_kpconv_fwd), with and
without the auxiliary streams of a forward."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superpoints_registration_amd import get_config, ops, synthetic, _lib
from superpoints_registration_amd.regtr import RegTR, no_side_stream
dev = torch.device('cuda:0')
cfg = get_config('3dmatch')
model = RegTR(cfg); synthetic.fill_parameters(model, 0); model = model.to(dev).eval()
model.inputs_resident = True
pairs = [synthetic.make_pair(16384, seed=i) for i in range(16)]
batch = {"src_xyz": [torch.from_numpy(p[0]).to(dev) for p in pairs], "tgt_xyz": [torch.from_numpy(p[1]).to(dev) for p in pairs]}
L = _lib.lib()
def run(tag, ctx):
    with ctx, torch.no_grad():
        for _ in range(3): model(dict(batch))
        torch.cuda.synchronize()
        L.spr_prof_enable(1)
        for _ in range(4): model(dict(batch))
        torch.cuda.synchronize()
    cap = 4096
    c, q, m = (ctypes.c_int * cap)(), (ctypes.c_int * cap)(), (ctypes.c_float * cap)()
    n = L.spr_prof_read(cap, c, q, m)
    L.spr_prof_enable(0)
    agg = {}
    for i in range(n):
        if c[i] > 0:
            agg.setdefault((c[i], q[i]), []).append(m[i])
    print(tag)
    for (code, nq), v in sorted(agg.items()):
        print('   %3d->%3d nq %7d  launches %2d  avg %.3f ms  min %.3f  max %.3f' % (code // 100000, code % 100000, nq, len(v), sum(v) / len(v), min(v), max(v)))
import contextlib
run('three streams (default)', contextlib.nullcontext())
run('one stream', no_side_stream())
