"""Pose / feature error of the small-weights case (tests/test_gpu_range.py) against the float64
CPU oracle, per arithmetic mode: which products eat the accuracy margin?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import torch_oracle as O
from superpoints_registration_amd import get_config, ops, synthetic
from superpoints_registration_amd.regtr import RegTR
dev = torch.device('cuda:0')
cfg = get_config("3dmatch")
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.01
src, tgt, _ = synthetic.make_pair(2048, seed=5, extent=0.6, jitter=0.002)
model = RegTR(cfg); synthetic.fill_parameters(model, seed=1)
with torch.no_grad():
    for name, p in model.named_parameters():
        if p.dim() >= 2 and not name.endswith(".W"):
            p.mul_(scale)
sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
ref = O.regtr_forward(cfg, {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}, [src], [tgt])
ref32 = O.regtr_forward(cfg, sd, [src], [tgt])
p64 = ref["pose"][0].numpy(); f64 = ref["cond"][0][0].numpy()
print('float32 CPU oracle: pose %.2e feat %.2e' % (np.linalg.norm(ref32["pose"][0].numpy() - p64), np.abs(ref32["cond"][0][0].numpy() - f64).max() / np.abs(f64).max()))
model = model.to(dev).eval()
T = torch.from_numpy
for gm, am in ((1, 1), (0, 1), (1, 0), (0, 0), (1, 2)):
    ops.set_gemm_mode(gm); ops.set_attn_mode(am)
    out = model({"src_xyz": [T(src).to(dev)], "tgt_xyz": [T(tgt).to(dev)]})
    pe = np.linalg.norm(out["pose"][0].cpu().numpy().astype(np.float64) - p64)
    sf = out["src_feat"][0][0].cpu().numpy().astype(np.float64)
    fe = np.abs(sf - f64.reshape(sf.shape)).max() / np.abs(f64).max()
    print('gemm_mode %d attn_mode %d: pose err %.3e  cond-feature err %.3e' % (gm, am, pe, fe))

# ---- pose head in isolation: the oracle's float32 conditioned features in, stage by stage ----
ops.set_gemm_mode(1); ops.set_attn_mode(1)
cond = torch.cat([ref32["cond"][0][0], ref32["cond"][1][0]] if isinstance(ref32["cond"], (list, tuple)) and len(ref32["cond"]) == 2 else
                 [ref32["cond"][0][0], ref32["tgt_cond"][0][0]]).float() if False else None
keys = list(ref32.keys()); print('oracle keys', keys)
