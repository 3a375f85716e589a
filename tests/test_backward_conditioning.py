"""CPU: conditioning of weight-gradient comparisons (context for tests/test_gpu_backward.py)."""
import numpy as np
import torch

from oracle import torch_oracle as O
from oracle.gen_golden import pairs_for
from superpoints_registration_amd import get_config, synthetic
from superpoints_registration_amd.regtr import RegTR

T = torch.from_numpy


def test_weight_gradients_are_discontinuous_in_the_activations():
    """Conditioning of the comparison above, inside the float64 oracle alone (no GPU arithmetic):
    an ABSOLUTE perturbation of 1e-5 of one block's output -- far below any tolerance of the
    forward -- moves individual weight-gradient entries of the following blocks by > 1e-4 of their
    scale (LeakyReLU branch flips), while a relative 1e-6 scaling, which cannot flip a sign, moves
    them by ~1e-6."""
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        cfg = get_config("modelnet")
        pairs, sizes = pairs_for("modelnet", 2)
        clouds = [p[0][:n] for p, (n, m) in zip(pairs, sizes)] + [p[1][:m] for p, (n, m) in zip(pairs, sizes)]
        model = RegTR(cfg)
        synthetic.fill_parameters(model, seed=0)
        sd0 = {k: (v.detach().clone().double() if v.is_floating_point() else v.clone()) for k, v in model.state_dict().items()}
        meta_np = O.preprocess(cfg, [np.asarray(c) for c in clouds])
        meta = {k: [T(np.ascontiguousarray(v)) for v in vs] for k, vs in meta_np.items()}
        meta["points"] = [p.double() for p in meta["points"]]
        lens, pts, idx = meta["stack_lengths"][1], meta["points"][1], meta["neighbors"][1]
        with torch.no_grad():
            _, feats = O.encoder(cfg, sd0, meta)
        x3 = feats[3].detach()
        G = synthetic.rand((x3.shape[0], 1024), 5).double()

        def grads(x_in):
            p = "kpf_encoder.encoder_blocks.4."
            sd = {k: v.clone().requires_grad_(True) for k, v in sd0.items() if k.startswith(p) and "kernel_points" not in k}
            ext = 2 * cfg.first_subsampling_dl * cfg.KP_extent      # level-1 extent (radius doubles once)
            t = O.unary(x_in, sd[p + "unary1.mlp.weight"], lens)
            t = O.lrelu(O.instance_norm(O.kpconv(pts, pts, idx, t, sd[p + "KPConv.weights"], sd0[p + "KPConv.kernel_points"], ext), lens))
            t = O.unary(t, sd[p + "unary2.mlp.weight"], lens, relu=False)
            sc = O.unary(x_in, sd[p + "unary_shortcut.mlp.weight"], lens, relu=False)
            (O.lrelu(t + sc) * G).sum().backward()
            return {k: v.grad for k, v in sd.items()}
        g0 = grads(x3)
        noise = synthetic.rand(tuple(x3.shape), 6).double()
        g_abs = grads(x3 + 1e-5 * noise)
        g_rel = grads(x3 * (1.0 + 1e-6 * noise))
        dev_abs = max(float((g_abs[k] - g0[k]).abs().max() / g0[k].abs().max()) for k in g0)
        dev_rel = max(float((g_rel[k] - g0[k]).abs().max() / g0[k].abs().max()) for k in g0)
        assert dev_rel < 2e-5, dev_rel
        assert dev_abs > 1e-4, dev_abs
    finally:
        torch.set_default_dtype(old)
