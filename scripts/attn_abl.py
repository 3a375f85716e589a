"""Times the attention core alone at bench-like shapes (ablation switch: SPR_ATTN_ABL)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superpoints_registration_amd import ops
dev = torch.device('cuda:0')
nseg, L, nhead = 32, 1930, 8
T = nseg * L
g = torch.Generator(device='cpu'); g.manual_seed(0)
qkv = torch.randn(T, 768, generator=g).to(dev)
cu = (torch.arange(0, nseg + 1, dtype=torch.int32) * L).to(dev)
kv = (torch.arange(nseg, dtype=torch.int32) ^ 1).to(dev)
out = torch.empty(T, 256, device=dev)
ops.set_attn_mode(int(os.environ.get('SPR_ATTN_MODE', '1')))
q, k, v = qkv[:, :256], qkv[:, 256:512], qkv[:, 512:]
for _ in range(3):
    ops.attention(q, k, v, cu, kv, L, nhead, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.attention(q, k, v, cu, kv, L, nhead, out=out)
e1.record(); torch.cuda.synchronize()
print('mode', os.environ.get('SPR_ATTN_MODE', '1'), 'us/call (range pre-pass + pack + core) %.1f' % (e0.elapsed_time(e1) * 100))
