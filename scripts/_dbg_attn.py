import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superpoints_registration_amd import ops
dev = torch.device('cuda:0')
for lens, kv_seg in (([1930, 701, 64, 2100], [1, 0, 3, 2]), ([700, 333], [0, 1]), ([128, 128], [0, 1]), ([64, 64], [1, 0])):
    tot = sum(lens)
    g = torch.Generator().manual_seed(23)
    q = torch.randn((tot, 256), generator=g); k = torch.randn((tot, 256), generator=g); v = torch.randn((tot, 256), generator=g)
    cu = ops.lengths_to_cu(lens, dev); seg = torch.tensor(kv_seg, dtype=torch.int32, device=dev)
    outs = {}
    for mode in (1, 3, 4):
        ops.set_attn_mode(mode)
        o = ops.attention(q.to(dev), k.to(dev), v.to(dev), cu, seg, max(lens), 8).cpu()
        outs[mode] = o
        bad = ~torch.isfinite(o)
        rows = bad.any(1).nonzero().flatten()
        print(lens, 'mode', mode, 'non-finite elements', int(bad.sum()), 'rows', rows[:8].tolist(), '...', rows[-4:].tolist() if len(rows) else [],
              'heads of first bad row', bad[rows[0]].view(8, 32).any(1).tolist() if len(rows) else None)
    print('  max |m4 - m1| on finite', float((outs[4] - outs[1]).nan_to_num(0).abs().max()))
ops.set_attn_mode(1)
