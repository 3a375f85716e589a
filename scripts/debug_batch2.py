import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle.gen_golden import pairs_for
from oracle import torch_oracle as O
from superpoints_registration_amd import get_config, ops, synthetic
from superpoints_registration_amd.regtr import RegTR

dev = torch.device('cuda:0')
cfg = get_config('3dmatch')
pairs, sizes = pairs_for('3dmatch', 2)
model = RegTR(cfg); synthetic.fill_parameters(model, 0); model = model.to(dev).eval()

def run(idx):
    src = [torch.from_numpy(pairs[b][0][:sizes[b][0]]).to(dev) for b in idx]
    tgt = [torch.from_numpy(pairs[b][1][:sizes[b][1]]).to(dev) for b in idx]
    meta = model.preprocessor(src + tgt)
    x = torch.ones((meta['points'][0].shape[0], 1), device=dev)
    blks = model.kpf_encoder.encoder_blocks
    x = blks[0](x, meta); x = blks[1](x, meta)
    blk = blks[2]
    feats = x
    li = 0
    lens0, lens1 = meta['_lens_host'][0], meta['_lens_host'][1]
    cu0, cu1 = meta['_cu'][0], meta['_cu'][1]
    pools = meta['_i32'][('pools', 0)]
    print('idx', idx, 'pools shape', tuple(pools.shape), 'stride', pools.stride(), 'lens', lens0, lens1)
    y1 = blk.unary1(feats, None, cu=cu0, max_len=max(lens0))
    r1 = O.unary(feats, blk.unary1.mlp.weight, lens0)
    print('  unary1 err', float((y1 - r1).abs().max()))
    k0 = blk.KPConv(meta['points'][1], meta['points'][0], pools, y1)
    blk.KPConv.impl = 1
    k1 = blk.KPConv(meta['points'][1], meta['points'][0], pools, y1)
    blk.KPConv.impl = 0
    kr = O.kpconv(meta['points'][1], meta['points'][0], pools.long(), y1, blk.KPConv.weights, blk.KPConv.kernel_points, blk.KPConv.KP_extent)
    print('  kpconv mfma err', float((k0 - kr).abs().max()), 'simple err', float((k1 - kr).abs().max()), 'absmax', float(kr.abs().max()))
    n1 = blk.batch_norm_conv(k0, None, cu=cu1, slope=0.1, max_len=max(lens1))
    nr = O.lrelu(O.instance_norm(k0, lens1))
    print('  norm err', float((n1 - nr).abs().max()))
    mp = ops.maxpool(feats, pools)
    mr = O.max_pool(feats, pools.long())
    print('  maxpool err', float((mp - mr).abs().max()))
    sr = mr
    out = blk(feats, meta)
    u2 = O.unary(nr, blk.unary2.mlp.weight, lens1, relu=False)
    outr = O.lrelu(u2 + sr)
    print('  block err', float((out - outr).abs().max()))

with torch.no_grad():
    run([0, 1]); run([1]); run([0])
