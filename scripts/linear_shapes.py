"""Lists every spr_linear call of one forward (shape, epilogue) with its GPU time."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superpoints_registration_amd import get_config, ops, synthetic
from superpoints_registration_amd.regtr import RegTR

dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
cfg = get_config('3dmatch')
model = RegTR(cfg); synthetic.fill_parameters(model, 0); model = model.to(dev).eval()
pairs = [synthetic.make_pair(16384, seed=i) for i in range(B)]
batch = {"src_xyz": [torch.from_numpy(p[0]).to(dev) for p in pairs],
         "tgt_xyz": [torch.from_numpy(p[1]).to(dev) for p in pairs]}
model(batch); torch.cuda.synchronize()
rec = []
orig = ops.linear
def timed(x, w, bias=None, residual=None, act=0):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = orig(x, w, bias, residual=residual, act=act); e1.record()
    rec.append(((x.shape[0], x.shape[1], w.shape[0], act, residual is not None), e0, e1))
    return r
ops.linear = timed
import superpoints_registration_amd.transformers as T, superpoints_registration_amd.kpconv_blocks as KB, superpoints_registration_amd.regtr as R
for m in (T, KB, R):
    if hasattr(m, 'ops'): m.ops.linear = timed
model(batch); torch.cuda.synchronize()
agg = collections.OrderedDict()
for key, e0, e1 in rec:
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1) * 1e3
tot = 0
for (m, k, n, act, res), (c, us) in agg.items():
    fl = 2.0 * m * k * n; by = 4.0 * (m * k + m * n * (2 if res else 1))
    print(f"m={m:7d} k={k:5d} n={n:5d} act={act} res={int(res)} calls={c:3d} us/call={us/c:8.1f}  TF={fl/(us/c)*1e-6:6.1f} GB/s={by/(us/c)*1e-3:7.0f}")
    tot += us
print('total us', tot)
