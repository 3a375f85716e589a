"""Times the attention core alone at bench-like shapes (32 clouds of 1 930 tokens = the bench's 16-pair call), per
attention mode (SPR_ATTN_MODES, default "1,3,2"); environment switches of the core (SPR_ATTN_CORE, SPR_ATTN_PRIO,
SPR_ATTN_NQ ...) are read once per process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superpoints_registration_amd import ops
dev = torch.device('cuda:0')
nseg, L, nhead = int(os.environ.get('NSEG', '32')), 1930, 8
T = nseg * L
g = torch.Generator(device='cpu'); g.manual_seed(0)
qkv = torch.randn(T, 768, generator=g).to(dev)
cu = (torch.arange(0, nseg + 1, dtype=torch.int32) * L).to(dev)
kv = (torch.arange(nseg, dtype=torch.int32) ^ 1).to(dev)
out = torch.empty(T, 256, device=dev)
q, k, v = qkv[:, :256], qkv[:, 256:512], qkv[:, 512:]
tag = ' '.join(f'{k_}={os.environ[k_]}' for k_ in ('SPR_ATTN_CORE', 'SPR_ATTN_PRIO', 'SPR_ATTN_NQ') if k_ in os.environ)
for mode in [int(m) for m in os.environ.get('SPR_ATTN_MODES', os.environ.get('SPR_ATTN_MODE', '1,4,3,2')).split(',')]:
    ops.set_attn_mode(mode)
    for _ in range(3):
        ops.attention(q, k, v, cu, kv, L, nhead, out=out)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.attention(q, k, v, cu, kv, L, nhead, out=out)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 100)
    # the core alone: HIP events inside the library around the core launch (what bench.py's roofline_attention reads)
    import ctypes
    from superpoints_registration_amd import _lib
    Lh = _lib.lib()
    Lh.spr_prof_enable(1)
    for _ in range(10):
        ops.attention(q, k, v, cu, kv, L, nhead, out=out)
    torch.cuda.synchronize()
    cap = 64
    codes, nqs, ms = (ctypes.c_int * cap)(), (ctypes.c_int * cap)(), (ctypes.c_float * cap)()
    n = Lh.spr_prof_read(cap, codes, nqs, ms)
    Lh.spr_prof_enable(0)
    core = [ms[i] for i in range(n) if codes[i] == -1]
    core_us = 1e3 * sum(core) / max(len(core), 1)
    flops = 4.0 * 256 * nseg * L * L
    print(f'[{tag}] mode {mode} us/call (range pre-pass + pack + core) {best:.1f}   core alone {core_us:.1f} us = '
          f'{flops / core_us / 1e6:.1f} TFLOP/s algorithmic = {flops / core_us / 1e6 / 2500:.4f} of 2.5 PFLOP/s')
ops.set_attn_mode(1)
