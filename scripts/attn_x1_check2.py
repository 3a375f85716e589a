import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superpoints_registration_amd import ops
dev = torch.device("cuda:0")
for n in [96, 100, 120, 126, 127, 128, 191, 256]:
    g = torch.Generator().manual_seed(n)
    q = torch.randn(n, 256, generator=g).to(dev); k = torch.randn(n, 256, generator=g).to(dev); v = torch.randn(n, 256, generator=g).to(dev)
    cu = ops.lengths_to_cu([n], dev); seg = torch.zeros(1, dtype=torch.int32, device=dev)
    o = ops.attention_raw(q, k, v, cu, seg, n, 8).cpu().double()
    qs, ks, vs = (t.cpu().double().view(n, 8, 32).transpose(0, 1) for t in (q, k, v))
    a = torch.softmax(qs @ ks.transpose(1, 2) / math.sqrt(32), -1)
    ref = (a @ vs).transpose(0, 1).reshape(n, 256)
    e = (o - ref).abs()
    perq = e.max(1).values
    blocks = [float(perq[i:i + 32].max()) for i in range(0, n, 32)]
    perh = [float(e[:, 32 * h:32 * h + 32].max()) for h in range(8)]
    # does the output equal attention over only the first 64 keys / only the last tile / all but ...?
    def part(lo, hi):
        a2 = torch.softmax(qs[:, :, :] @ ks[:, lo:hi].transpose(1, 2) / math.sqrt(32), -1)
        return (a2 @ vs[:, lo:hi]).transpose(0, 1).reshape(n, 256)
    alt = {"first64": float((o - part(0, 64)).abs().max()), "from64": float((o - part(64, n)).abs().max()) if n > 64 else -1}
    print(n, "blocks", " ".join(f"{b:.1e}" for b in blocks), "| heads", " ".join(f"{b:.1e}" for b in perh), "|", alt)
