"""Pose-head error budget in the small-weights case: GPU (w, t_hat) vs float64 from the SAME
conditioned features, and the Procrustes solve on each."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from oracle import torch_oracle as O
from superpoints_registration_amd import get_config, ops, synthetic
from superpoints_registration_amd.regtr import RegTR
dev = torch.device('cuda:0')
cfg = get_config("3dmatch")
src, tgt, _ = synthetic.make_pair(2048, seed=5, extent=0.6, jitter=0.002)
model = RegTR(cfg); synthetic.fill_parameters(model, seed=1)
with torch.no_grad():
    for name, p in model.named_parameters():
        if p.dim() >= 2 and not name.endswith(".W"):
            p.mul_(0.01)
model = model.to(dev).eval()
T = torch.from_numpy
out = model({"src_xyz": [T(src).to(dev)], "tgt_xyz": [T(tgt).to(dev)]})
cs, ct = out["src_feat"][0][0], out["tgt_feat"][0][0]
sx, tx = out["src_kp"][0], out["tgt_kp"][0]
n, m = cs.shape[0], ct.shape[0]
cond = torch.cat([cs, ct]); xyz = torch.cat([sx, tx])
cu = torch.tensor([0, n, n + m], dtype=torch.int32, device=dev)
w, that = ops.sinkhorn_correspondences(cond, xyz, cu, [0, n, n + m], 1, model.alpha, model.beta, int(cfg.sinkhorn_itr), bool(cfg.slack))
pose_gpu = ops.weighted_procrustes(sx, that, w, cu[:2].contiguous())[0].double().cpu()
c64, t64 = cs.double().cpu(), ct.double().cpu(); x64s, x64t = sx.double().cpu(), tx.double().cpu()
score = torch.clamp(c64 @ t64.t() / math.sqrt(c64.shape[1]), min=0.0)
aff = -(score - F.softplus(model.alpha.detach().double().cpu())) / (math.exp(float(model.beta)) + 0.02)
la = F.pad(aff, (0, 1, 0, 1))
for _ in range(cfg.sinkhorn_itr):
    la = torch.cat((la[:-1] - torch.logsumexp(la[:-1], dim=1, keepdim=True), la[-1:]), 0)
    la = torch.cat((la[:, :-1] - torch.logsumexp(la[:, :-1], dim=0, keepdim=True), la[:, -1:]), 1)
perm = torch.exp(la[:-1, :-1]); w64 = perm.sum(1); th64 = perm @ x64t / (w64[:, None] + 1e-6)
p64 = O.compute_rigid_transform(x64s, th64, w64).double()
print('n, m', n, m, 'w range', float(w64.min()), float(w64.max()), 'aff range', float(aff.min()), float(aff.max()))
print('w rel err max', float(((w.double().cpu() - w64) / w64).abs().max()))
spread = float((th64 - th64.mean(0)).abs().max())
print('t_hat abs err max', float((that.double().cpu() - th64).abs().max()), 'spread', spread)
print('pose: gpu head + gpu procrustes vs f64', float((pose_gpu - p64).norm()))
print('pose: gpu (w,t_hat) + f64 procrustes vs f64', float((O.compute_rigid_transform(x64s, that.double().cpu(), w.double().cpu()).double() - p64).norm()))
print('pose: f64 (w,t_hat) rounded to f32 + gpu procrustes', float((ops.weighted_procrustes(sx, th64.float().to(dev), w64.float().to(dev), cu[:2].contiguous())[0].double().cpu() - p64).norm()))
print('pose: exact w, gpu t_hat', float((O.compute_rigid_transform(x64s, that.double().cpu(), w64).double() - p64).norm()))
print('pose: gpu w, exact t_hat', float((O.compute_rigid_transform(x64s, th64, w.double().cpu()).double() - p64).norm()))
