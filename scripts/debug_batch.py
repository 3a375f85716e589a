import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle.gen_golden import pairs_for
from superpoints_registration_amd import get_config, ops, synthetic
from superpoints_registration_amd.regtr import RegTR
from superpoints_registration_amd.transformers import make_segments

dev = torch.device('cuda:0')
cfg = get_config('3dmatch')
pairs, sizes = pairs_for('3dmatch', 2)
model = RegTR(cfg); synthetic.fill_parameters(model, 0); model = model.to(dev).eval()

def run(idx):
    src = [torch.from_numpy(pairs[b][0][:sizes[b][0]]).to(dev) for b in idx]
    tgt = [torch.from_numpy(pairs[b][1][:sizes[b][1]]).to(dev) for b in idx]
    B = len(idx)
    meta = model.preprocessor(src + tgt)
    feats0 = torch.ones((meta['points'][0].shape[0], 1), device=dev)
    outs = []
    x = feats0
    for blk in model.kpf_encoder.encoder_blocks:
        x = blk(x, meta)
        outs.append(x)
    return meta, outs

m2, o2 = run([0, 1])
m1, o1 = run([1])
# clouds order for B=2: [s0, s1, t0, t1]; for B=1: [s1, t1]
for l in range(len(m2['points'])):
    L2 = m2['_lens_host'][l]; L1 = m1['_lens_host'][l]
    c2 = np.concatenate([[0], np.cumsum(L2)]); c1 = np.concatenate([[0], np.cumsum(L1)])
    print('level', l, L2, L1)
    for (a, b) in ((1, 0), (3, 1)):
        p2 = m2['points'][l][c2[a]:c2[a+1]]; p1 = m1['points'][l][c1[b]:c1[b+1]]
        print('  points equal', torch.equal(p2, p1))
for i, (a, b) in enumerate(zip(o2, o1)):
    blk = model.kpf_encoder.encoder_blocks[i]
    l = blk.layer_ind + (1 if 'strided' in blk.block_name else 0)
    L2 = m2['_lens_host'][l]; L1 = m1['_lens_host'][l]
    c2 = np.concatenate([[0], np.cumsum(L2)]); c1 = np.concatenate([[0], np.cumsum(L1)])
    d = 0.0
    for (u, v) in ((1, 0), (3, 1)):
        d = max(d, float((a[c2[u]:c2[u+1]] - b[c1[v]:c1[v+1]]).abs().max()))
    print('block', i, blk.block_name, 'max abs diff', d, 'absmax', float(a.abs().max()))
# repeatability of same call
m2b, o2b = run([0, 1])
print('repeat identical:', all(torch.equal(a, b) for a, b in zip(o2, o2b)))
