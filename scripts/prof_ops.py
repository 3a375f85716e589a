"""Runs the three heavy operators at bench sizes a few times (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superpoints_registration_amd import get_config, ops, synthetic
from superpoints_registration_amd.regtr import RegTR

dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfg = get_config('3dmatch')
model = RegTR(cfg); synthetic.fill_parameters(model, 0); model = model.to(dev).eval()
pairs = [synthetic.make_pair(16384, seed=i) for i in range(B)]
batch = {"src_xyz": [torch.from_numpy(p[0]).to(dev) for p in pairs],
         "tgt_xyz": [torch.from_numpy(p[1]).to(dev) for p in pairs]}
out = model(batch)
torch.cuda.synchronize()
meta = batch['kpconv_meta']
print('levels', [tuple(p.shape) for p in meta['points']], [tuple(n.shape) for n in meta['neighbors']])
for _ in range(reps):
    out = model(batch)
torch.cuda.synchronize()
print('done')
