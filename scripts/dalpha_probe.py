"""Round 5: where do the Sinkhorn scalars' gradients (d alpha, d beta) stand?  Ours (GPU, fp32 arithmetic), the
reference's own fp32 backward (golden fixture) and the float64 CPU oracle's autograd on the same step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from conftest import load_golden
from oracle import torch_oracle as O
from oracle.gen_golden import loss_inputs, pairs_for
from superpoints_registration_amd import get_config, synthetic
from superpoints_registration_amd.regtr import RegTR
T = torch.from_numpy
dev = torch.device('cuda:0')
for tag in ("3dmatch", "kitti", "modelnet"):
    g = load_golden(f"grad_{tag}_b2.npz")
    B = int(g["B"]); cfg = get_config(tag)
    if not cfg.use_sinkhorn:
        print(tag, "no sinkhorn"); continue
    pairs, sizes = pairs_for(tag, B)
    pose, src_ov, tgt_ov = loss_inputs(tag, B)
    model = RegTR(cfg); synthetic.fill_parameters(model, seed=int(g["seed"]))
    sd64 = {k: (v.detach().clone().double() if v.is_floating_point() else v.clone()) for k, v in model.state_dict().items()}
    for k in ("alpha", "beta"):
        sd64[k].requires_grad_(True)
    src = [p[0][:n] for p, (n, m) in zip(pairs, sizes)]; tgt = [p[1][:m] for p, (n, m) in zip(pairs, sizes)]
    fwd = O.regtr_forward(cfg, sd64, src, tgt)
    L = O.compute_loss(cfg, sd64, fwd, pose, src_ov, tgt_ov)
    L["total"].backward()
    model = model.to(dev).train()
    batch = {"src_xyz": [T(s).to(dev) for s in src], "tgt_xyz": [T(t).to(dev) for t in tgt], "pose": T(pose).to(dev),
             "src_overlap": [T(o).to(dev) for o in src_ov], "tgt_overlap": [T(o).to(dev) for o in tgt_ov]}
    out = model(batch); losses = model.compute_loss(out, batch)
    model.zero_grad(set_to_none=True); losses["total"].backward()
    for k in ("alpha", "beta"):
        ours = float(getattr(model, k).grad); ref = float(np.asarray(g[f"total|{k}|full"]).reshape(-1)[0]); f64 = float(sd64[k].grad)
        print(f"{tag} d{k}: ours {ours:.9e}  reference fp32 {ref:.9e}  float64 oracle {f64:.9e} | ours vs f64 {abs(ours-f64)/abs(f64):.2e}  ref vs f64 {abs(ref-f64)/abs(f64):.2e}  ours vs ref {abs(ours-ref)/abs(ref):.2e}")
