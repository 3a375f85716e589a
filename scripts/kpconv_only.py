"""Runs the 64->64 KPConv of level 1 a few times (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superpoints_registration_amd import get_config, ops, synthetic
from superpoints_registration_amd.regtr import RegTR
dev = torch.device('cuda:0')
cfg = get_config('3dmatch')
model = RegTR(cfg); synthetic.fill_parameters(model, 0); model = model.to(dev).eval()
pairs = [synthetic.make_pair(16384, seed=i) for i in range(16)]
meta = model.preprocessor([torch.from_numpy(p[0]).to(dev) for p in pairs] + [torch.from_numpy(p[1]).to(dev) for p in pairs])
pts = meta['points'][1]; nb = meta['_i32'][('neighbors', 1)]
blk = model.kpf_encoder.encoder_blocks[3].KPConv
x = torch.rand((pts.shape[0], 64), device=dev) - 0.3
for _ in range(4):
    ops.kpconv(pts, pts, nb, x, blk.weights.detach(), blk.kernel_points.detach(), blk.KP_extent, rows_sorted=True)
torch.cuda.synchronize()
