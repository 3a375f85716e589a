"""Per-segment error of the attention core against a float64 softmax(QK^T/sqrt(d))V for a list of segment-length
sets (A/B of attention kernels: run once with SPR_ATTN_X1=1 and once with 0)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superpoints_registration_amd import ops

dev = torch.device("cuda:0")
cases = [[128, 128], [64, 64], [128], [192], [256], [320], [70, 129], [1930, 1800], [127], [129], [63, 65], [512]]
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
for lens in cases:
    T = sum(lens)
    g = torch.Generator().manual_seed(T)
    q = (torch.randn(T, 256, generator=g) * scale).to(dev)
    k = (torch.randn(T, 256, generator=g) * scale).to(dev)
    v = torch.randn(T, 256, generator=g).to(dev)
    cu = ops.lengths_to_cu(lens, dev)
    seg = torch.arange(len(lens), dtype=torch.int32, device=dev)
    o = ops.attention_raw(q, k, v, cu, seg, max(lens), 8).cpu().double()
    errs = []
    off = 0
    for n in lens:
        qs, ks, vs = (t[off:off + n].cpu().double().view(n, 8, 32).transpose(0, 1) for t in (q, k, v))
        a = torch.softmax(qs @ ks.transpose(1, 2) / math.sqrt(32), -1)
        ref = (a @ vs).transpose(0, 1).reshape(n, 256)
        errs.append(float((o[off:off + n] - ref).abs().max() / ref.abs().max()))
        off += n
    print(lens, " ".join(f"{e:.2e}" for e in errs))
