"""GPU: the fused cross-encoder stack (csrc/xenc.hip, spr_xenc_forward) against the reference's
golden layer output, the float64 oracle of the same layers (oracle/torch_oracle.py: layer_pre /
transformer, which restate transformers.py:184-245 and :27-59) and the per-operator HIP route."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import torch_oracle as O
from oracle.gen_golden import ops_inputs
from superpoints_registration_amd import get_config, ops, synthetic
from superpoints_registration_amd.transformers import (TransformerCrossEncoder, TransformerCrossEncoderLayer,
                                                       make_segments)

pytestmark = pytest.mark.gpu


def _close(got, ref, rel, what=""):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    err = np.abs(got - ref).max()
    scale = max(np.abs(ref).max(), 1e-30)
    assert err <= rel * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e} (rel {err / scale:.3e} > {rel})"
    return err / scale


def _encoder(device, n_layers, d_ff=1024, final=True, seed=21):
    layer = TransformerCrossEncoderLayer(256, 8, d_ff, 0.0, 'relu', True, True, True, 'dot_prod')
    enc = TransformerCrossEncoder(layer, n_layers, torch.nn.LayerNorm(256) if final else None)
    synthetic.fill_parameters(enc, seed=seed)
    return enc.to(device)


def _oracle64(enc, x, pos, s_l, t_l):
    """float64 CPU evaluation of the stack, pair by pair (the oracle works on one unpadded pair)."""
    sd = {k: v.detach().cpu().double() for k, v in enc.state_dict().items()}
    cfg = get_config("3dmatch")
    cfg.num_encoder_layers = enc.num_layers
    x, pos = x.cpu().double(), pos.cpu().double()
    B = len(s_l)
    off = np.concatenate([[0], np.cumsum(list(s_l) + list(t_l))])
    out = torch.empty_like(x)
    for b in range(B):
        s = slice(off[b], off[b + 1])
        t = slice(off[B + b], off[B + b + 1])
        if enc.norm is not None:
            so, to = O.transformer(cfg, sd, x[s], x[t], pos[s], pos[t], prefix='')
        else:
            so, to = x[s], x[t]
            for l in range(enc.num_layers):
                so, to = O.layer_pre(sd, f'layers.{l}.', so, to, pos[s], pos[t])
        out[s], out[t] = so, to
    return out.numpy()


def test_one_layer_against_the_reference_golden(device):
    gold, inp = load_golden("ops.npz"), ops_inputs()
    enc = TransformerCrossEncoder(TransformerCrossEncoderLayer(256, 8, 1024, 0.0, 'relu', True, True, True, 'dot_prod'), 1, None)
    synthetic.fill_parameters(enc.layers[0], seed=21)
    enc = enc.to(device)
    x = torch.cat(inp["tl.src"] + inp["tl.tgt"]).to(device)
    pe = torch.cat(inp["tl.src_pe"] + inp["tl.tgt_pe"]).to(device)
    cu, s_self, s_cross, mx = make_segments(inp["tl.s_l"], inp["tl.t_l"], device)
    ns = sum(inp["tl.s_l"])
    with torch.no_grad():
        y = enc.forward_packed(x, cu, s_self, s_cross, mx, pos=pe, pos_bound=float(pe.abs().max())).cpu().numpy()
    assert getattr(enc, '_spr_xenc', None) is not None, "the fused route was not taken"
    _close(y[:ns], gold["tl.src_out"], 2e-5, "fused layer src")     # same bound as the per-operator layer test
    _close(y[ns:], gold["tl.tgt_out"], 2e-5, "fused layer tgt")


@pytest.mark.parametrize("n_layers,d_ff,final,s_l,t_l", [
    (2, 1024, True, [70, 129, 33], [200, 1, 64]),          # ragged, a one-token cloud, last tile partly empty
    (3, 64, False, [128, 128], [128, 128]),                # whole tiles, two hidden chunks, no final norm
    (6, 1024, True, [1930, 1800], [1711, 2048]),           # the bench's superpoint counts, full depth
])
def test_stack_against_float64_and_the_operator_route(device, n_layers, d_ff, final, s_l, t_l):
    enc = _encoder(device, n_layers, d_ff, final)
    g = torch.Generator().manual_seed(5)
    T = sum(s_l) + sum(t_l)
    x = (torch.randn(T, 256, generator=g) * 1.7).to(device)
    pos = torch.rand(T, 256, generator=g).mul(2).sub(1).to(device)          # |pos| <= 1 like the sine embedding
    cu, s_self, s_cross, mx = make_segments(s_l, t_l, device)
    with torch.no_grad():
        fused = enc.forward_packed(x, cu, s_self, s_cross, mx, pos=pos, pos_bound=1.0)
        assert getattr(enc, '_spr_xenc', None) is not None
        plain = enc.forward_packed(x, cu, s_self, s_cross, mx, pos=pos)     # no bound given: operator by operator
        again = enc.forward_packed(x, cu, s_self, s_cross, mx, pos=pos, pos_bound=1.0)
    assert torch.equal(fused, again), "fused stack is not deterministic"
    ref = _oracle64(enc, x, pos, s_l, t_l)
    e_f = _close(fused.cpu().numpy(), ref, 2e-5, "fused stack vs float64")          # layer-test bound
    e_p = _close(plain.cpu().numpy(), ref, 2e-5, "operator route vs float64")
    # the fused chains must not be less accurate than the operators they replace (both ~1e-6)
    assert e_f <= max(3 * e_p, 3e-6), (e_f, e_p)


def test_many_tiles_per_workgroup_and_plan_refresh(device):
    """More tiles than workgroups (the ring wraps from one tile's stream into the next), and a parameter
    update invalidates the prepared weights."""
    enc = _encoder(device, 1, 64, True)
    g = torch.Generator().manual_seed(9)
    s_l, t_l = [9000, 9500, 8000], [9100, 7000, 9900]                       # 52 500 tokens = 411 tiles
    T = sum(s_l) + sum(t_l)
    x = torch.randn(T, 256, generator=g).to(device)
    pos = torch.rand(T, 256, generator=g).mul(2).sub(1).to(device)
    cu, s_self, s_cross, mx = make_segments(s_l, t_l, device)
    with torch.no_grad():
        fused = enc.forward_packed(x, cu, s_self, s_cross, mx, pos=pos, pos_bound=1.0)
        plain = enc.forward_packed(x, cu, s_self, s_cross, mx, pos=pos)
    _close(fused.cpu().numpy(), plain.cpu().numpy(), 5e-6, "fused vs operator route, 411 tiles")
    plan0 = enc._spr_xenc
    with torch.no_grad():
        enc.layers[0].linear1.weight.mul_(1.5)                               # bumps the version counter
        fused2 = enc.forward_packed(x, cu, s_self, s_cross, mx, pos=pos, pos_bound=1.0)
        plain2 = enc.forward_packed(x, cu, s_self, s_cross, mx, pos=pos)
    assert enc._spr_xenc is not plan0
    _close(fused2.cpu().numpy(), plain2.cpu().numpy(), 5e-6, "after the weight update")
    assert not torch.equal(fused, fused2)


def test_operand_magnitudes(device):
    """Static bounds instead of measured ranges: tokens and parameters at 1e-6 .. 1e+6 of the usual scale."""
    for xs, ws in ((1e-6, 1.0), (1e6, 1.0), (1.0, 1e-3), (1.0, 30.0)):
        enc = _encoder(device, 2, 1024, True, seed=3)
        with torch.no_grad():
            for l in enc.layers:
                for w in (l.linear1.weight, l.linear2.weight, l.self_attn.out_proj.weight, l.multihead_attn.out_proj.weight):
                    w.mul_(ws)
        g = torch.Generator().manual_seed(1)
        s_l, t_l = [300, 200], [250, 310]
        T = sum(s_l) + sum(t_l)
        x = (torch.randn(T, 256, generator=g) * xs).to(device)
        pos = torch.rand(T, 256, generator=g).mul(2).sub(1).to(device)
        cu, s_self, s_cross, mx = make_segments(s_l, t_l, device)
        with torch.no_grad():
            fused = enc.forward_packed(x, cu, s_self, s_cross, mx, pos=pos, pos_bound=1.0)
        ref = _oracle64(enc, x, pos, s_l, t_l)
        _close(fused.cpu().numpy(), ref, 2e-5, f"fused stack, x scale {xs}, w scale {ws}")


def test_many_small_segments(device):
    """Hundreds of tiny clouds (one tile each, most of it empty) through the fused route; beyond the LDS tables'
    capacity (cu_seqlens + tile prefix live behind the weight ring) the stack falls back to the operators."""
    enc = _encoder(device, 1, 64, True)
    g = torch.Generator().manual_seed(13)
    for npairs, fused_expected in ((300, True), (1500, False)):
        s_l = [int(v) for v in torch.randint(1, 9, (npairs,), generator=g)]
        t_l = [int(v) for v in torch.randint(1, 9, (npairs,), generator=g)]
        T = sum(s_l) + sum(t_l)
        x = torch.randn(T, 256, generator=g).to(device)
        pos = torch.rand(T, 256, generator=g).mul(2).sub(1).to(device)
        cu, s_self, s_cross, mx = make_segments(s_l, t_l, device)
        enc._spr_xenc = None
        with torch.no_grad():
            y = enc.forward_packed(x, cu, s_self, s_cross, mx, pos=pos, pos_bound=1.0)
            assert (enc._spr_xenc is not None) == fused_expected
            plain = enc.forward_packed(x, cu, s_self, s_cross, mx, pos=pos)
        _close(y.cpu().numpy(), plain.cpu().numpy(), 5e-6, f"{2 * npairs} segments")


def test_row_major_attention_output_route(device, tmp_path):
    """The chains read the attention output TILED when the core that runs is k_attn_s (it writes the tiles itself)
    and row-major otherwise (SPR_ATTN_CORE=h3: the round-4 core).  Both routes, each in its own process (the switch is
    read once), must agree to rounding: ragged clouds incl. a one-token and a 33-token one."""
    import subprocess, sys
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_xenc_route_child.py")
    outs = {}
    for core in ("s", "h3"):
        env = dict(os.environ)
        env.pop("SPR_ATTN_CORE", None)
        if core == "h3":
            env["SPR_ATTN_CORE"] = "h3"
        f = str(tmp_path / f"route_{core}.pt")
        subprocess.run([sys.executable, child, f], check=True, env=env, timeout=300)
        outs[core] = torch.load(f)
    scale = float(outs["s"].abs().max())
    assert torch.isfinite(outs["s"]).all() and torch.isfinite(outs["h3"]).all()
    assert float((outs["s"] - outs["h3"]).abs().max()) <= 5e-6 * scale
