import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
from oracle import torch_oracle as O
from superpoints_registration_amd import get_config, ops, synthetic
from superpoints_registration_amd.regtr import RegTR
T = torch.from_numpy
device = torch.device('cuda')
tag = '3dmatch'
cfg = get_config(tag)
model = RegTR(cfg); synthetic.fill_parameters(model, seed=0); model = model.to(device).train()
src, tgt, _ = synthetic.make_pair(16384, seed=0)
pts = [T(src).to(device), T(tgt).to(device)]
meta = model.preprocessor(pts)
enc = model.kpf_encoder
acts, hooks, block_in, ours_g = [], [], [], {}
for bi, blk in enumerate(enc.encoder_blocks):
    if hasattr(blk, 'unary1') and not isinstance(blk.unary1, torch.nn.Identity):
        hooks.append(blk.unary1.register_forward_hook(lambda m, a, o: acts.append(o.detach())))
    if hasattr(blk, 'batch_norm_conv'):
        hooks.append(blk.batch_norm_conv.register_forward_hook(lambda m, a, o: acts.append(o.detach())))
    def out_hook(m, a, o, bi=bi):
        acts.append(o.detach())
        if bi < 3:
            o.register_hook(lambda g, bi=bi: ours_g.__setitem__(('out', bi), g.detach().clone()))
    hooks.append(blk.register_forward_hook(out_hook))
    hooks.append(blk.register_forward_pre_hook(lambda m, a: block_in.append(a[0].detach())))
    if bi == 0:
        def conv_hook(m, a, o):
            o.register_hook(lambda g: ours_g.__setitem__('conv0', g.detach().clone()))
        hooks.append(blk.KPConv.register_forward_hook(conv_hook))
x0 = torch.ones((meta['points'][0].shape[0], 1), device=device)
f, _ = enc(x0, meta)
G = synthetic.rand(tuple(f.shape), 77, -1.0, 1.0).to(device)
enc.zero_grad(set_to_none=True)
(f * G).sum().backward()
meta64 = {'points': [p.double().cpu() for p in meta['points']], 'stack_lengths': [l.cpu() for l in meta['stack_lengths']],
          'neighbors': [n.long().cpu() for n in meta['neighbors']], 'pools': [n.long().cpu() for n in meta['pools']]}
pool_args = []
for i, name in enumerate(cfg.architecture):
    if 'strided' in name and 'resnetb' in name:
        lvl = sum(1 for n in cfg.architecture[:i] if 'strided' in n or 'pool' in n)
        xin = block_in[i].cpu(); x_ext = torch.cat((xin, torch.zeros_like(xin[:1])), 0)
        pool_args.append(x_ext[meta64['pools'][lvl]].max(1)[1])
sd = {k: v.detach().double().cpu().requires_grad_(v.requires_grad and 'kernel_points' not in k)
      for k, v in model.state_dict(keep_vars=True).items() if k.startswith('kpf_encoder.')}
frozen = O.FrozenDecisions([a.cpu() > 0 for a in acts], pool_args)
f64, feats64 = O.encoder(cfg, sd, meta64, frozen=frozen)
ref_g = {}
for bi in range(3):
    feats64[bi].register_hook(lambda g, bi=bi: ref_g.__setitem__(('out', bi), g.detach().clone()))
(f64 * G.double().cpu()).sum().backward()
lens = meta64['stack_lengths'][0]
for bi in range(3):
    a, b = ours_g[('out', bi)].double().cpu(), ref_g[('out', bi)]
    d = a - b
    off = 0; cm = []
    for n in [int(v) for v in meta64['stack_lengths'][0 if bi < 3 else 1]][:2]:
        cm.append(float(d[off:off + n].mean(0).abs().max())); off += n
    print('grad at block %d output: max err %.2e of max %.2e; rms err %.2e; per-cloud column-mean of the error (common mode) max %.2e; column mean of grad itself %.2e' % (
        bi, float(d.abs().max()), float(b.abs().max()), float(d.pow(2).mean().sqrt()), max(cm), float(b.mean(0).abs().max())))
name = 'kpf_encoder.encoder_blocks.0.KPConv.weights'
gw, rw = dict(model.named_parameters())[name].grad.double().cpu(), sd[name].grad
print('dW0 err %.2e of scale %.2e' % (float((gw - rw).abs().max()), float(rw.abs().max())))
print('ratio ours/ref (first kernel points, channel 0..3):', (gw[:4, 0, :4] / rw[:4, 0, :4]).numpy().round(5).tolist())
# ---- block 0 in isolation, float64, driven by the ORACLE's own gradient at the block output
W0 = sd['kpf_encoder.encoder_blocks.0.KPConv.weights'].detach().clone().requires_grad_(True)
KP0 = sd['kpf_encoder.encoder_blocks.0.KPConv.kernel_points'].detach()
ext = cfg.first_subsampling_dl * cfg.KP_extent
p0 = meta64['points'][0]
y64 = O.kpconv(p0, p0, meta64['neighbors'][0], torch.ones((p0.shape[0], 1), dtype=torch.float64), W0, KP0, ext)
y64.retain_grad()
mask0 = acts[0].cpu() > 0
z64 = O.instance_norm(y64, lens) * torch.where(mask0, torch.ones_like(y64), torch.full_like(y64, 0.1))
z64.backward(ref_g[('out', 0)])
a, b = ours_g['conv0'].double().cpu(), y64.grad
d = a - b
off = 0
for n in [int(v) for v in lens]:
    print('cloud of %d: dL/dy err max %.2e rms %.2e (of max %.2e); column SUM of ours %.3e, of oracle %.3e, of the error %.3e' % (
        n, float(d[off:off+n].abs().max()), float(d[off:off+n].pow(2).mean().sqrt()), float(b.abs().max()),
        float(a[off:off+n].sum(0).abs().max()), float(b[off:off+n].sum(0).abs().max()), float(d[off:off+n].sum(0).abs().max())))
    off += n
print('isolated float64 dW0 vs full-oracle dW0: %.2e' % float((W0.grad - rw).abs().max()))
# our dW recomputed in float64 from OUR dL/dy and the float64 weighted features: which input carries the error?
nb0 = meta64['neighbors'][0]
s_ext = torch.cat((p0, torch.full_like(p0[:1], 1e6)), 0)
nbp = s_ext[nb0] - p0.unsqueeze(1)
w = torch.clamp(1 - torch.sqrt(((nbp.unsqueeze(2) - KP0) ** 2).sum(3)) / ext, min=0.0).transpose(1, 2)   # [N, 15, K]
valid = (nb0 < p0.shape[0]).double()
wf64 = (w * valid.unsqueeze(1)).sum(2)                      # x == 1
cnt64 = valid.sum(1).clamp(min=1)
dW_a = (wf64 / cnt64.unsqueeze(1)).t() @ a                  # ours dL/dy, float64 everything else
dW_b = (wf64 / cnt64.unsqueeze(1)).t() @ b
print('float64 product with OUR dL/dy: err %.2e; with the oracle dL/dy: err %.2e (vs full-oracle dW0, scale %.2e)' % (
    float((dW_a - rw[:, 0, :]).abs().max()), float((dW_b - rw[:, 0, :]).abs().max()), float(rw.abs().max())))
