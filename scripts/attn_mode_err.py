"""Round 5: what the single-probability-plane attention mode (spr_set_attn_mode(3)) costs in accuracy, end to end, on
BASELINE-size pairs: conditioned features and pose of modes 1 / 3 / 2 against the exact-f32 forward (mode 0), for the
bench's random weights and for sharpened attention (q / k projection rows scaled: peaked softmax rows, few effective
keys per query -- the regime where per-probability rounding does not average out)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from superpoints_registration_amd import get_config, ops, synthetic
from superpoints_registration_amd.regtr import RegTR
dev = torch.device('cuda:0')
cfg = get_config("3dmatch")
npts = int(os.environ.get("POINTS", "16384"))
pairs = [synthetic.make_pair(npts, seed=70 + i) for i in range(2)]
batch = lambda: {"src_xyz": [torch.from_numpy(p[0]).to(dev) for p in pairs],
                 "tgt_xyz": [torch.from_numpy(p[1]).to(dev) for p in pairs]}
for sharpen in (1.0, 3.0, 8.0):
    model = RegTR(cfg); synthetic.fill_parameters(model, seed=0)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("in_proj_weight"):
                p[:512].mul_(sharpen ** 0.5)          # q and k rows: scores x sharpen
    model = model.to(dev).eval()
    outs = {}
    with torch.no_grad():
        for gm, am in ((0, 0), (1, 1), (1, 4), (1, 3), (1, 2)):
            ops.set_gemm_mode(gm); ops.set_attn_mode(am)
            outs[(gm, am)] = model(batch())
    ops.set_gemm_mode(1); ops.set_attn_mode(1)
    ex = outs[(0, 0)]
    for key in ((1, 1), (1, 4), (1, 3), (1, 2)):
        o = outs[key]
        fe = max(float((o["src_feat"][b][0] - ex["src_feat"][b][0]).abs().max()) / float(ex["src_feat"][b][0].abs().max())
                 for b in range(2))
        pe = max(float(np.linalg.norm(o["pose"][b].cpu().numpy() - ex["pose"][b].cpu().numpy())) for b in range(2))
        print(f"sharpen {sharpen:4.1f}  gemm/attn mode {key}: features {fe:.3e} of scale, pose {pe:.3e} (vs exact f32)")
