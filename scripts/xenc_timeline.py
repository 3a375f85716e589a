"""Per-stage time line of the fused cross-encoder chains (wave 0 of workgroup 0): needs the diagnostic build
of the library,
    make -C superpoints_registration_amd/csrc OUT=../libspr_hip_stamp.so BUILD=build/stamp/ EXTRA=-DSPR_XENC_STAMP
    SPR_HIP_LIB=superpoints_registration_amd/libspr_hip_stamp.so python scripts/xenc_timeline.py [tokens] [final|nofinal] [layers]
Every chain launch overwrites the stamps, so the table is that of the stack's LAST launch: with `final` chain B of
the last layer (feed-forward + final norm), with `nofinal 1` ... the same without a tail; to see a chain with the
in-projection tail run `tokens nofinal 2 A`, which stops the stamps after the first chain A (SPR_XENC_STAMP_STOP)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from superpoints_registration_amd import _lib, ops, synthetic
from superpoints_registration_amd.transformers import TransformerCrossEncoder, TransformerCrossEncoderLayer, make_segments

T = int(sys.argv[1]) if len(sys.argv) > 1 else 123500
final = (sys.argv[2] != "nofinal") if len(sys.argv) > 2 else True
nl = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = torch.device("cuda:0")
layer = TransformerCrossEncoderLayer(256, 8, 1024, 0.0, 'relu', True, True, True, 'dot_prod')
enc = TransformerCrossEncoder(layer, nl, torch.nn.LayerNorm(256) if final else None)
synthetic.fill_parameters(enc, seed=1)
enc = enc.to(dev)
n = T // 64
s_l, t_l = [n] * 32, [n] * 32
T = 64 * n
g = torch.Generator().manual_seed(0)
x = torch.randn(T, 256, generator=g).to(dev)
pos = torch.rand(T, 256, generator=g).mul(2).sub(1).to(dev)
cu, ss, sc, mx = make_segments(s_l, t_l, dev)
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * (64 * 64))()
with torch.no_grad():
    for _ in range(3):
        y = enc.forward_packed(x, cu, ss, sc, mx, pos=pos, pos_bound=1.0)
    torch.cuda.synchronize()
    raw.spr_xenc_debug_stamps(buf, 1)
    y = enc.forward_packed(x, cu, ss, sc, mx, pos=pos, pos_bound=1.0)
    torch.cuda.synchronize()
raw.spr_xenc_debug_stamps(buf, 0)
st = np.array(buf[:], dtype=np.uint64).reshape(64, 64).astype(np.int64)
names = {0: "tile start", 1: "o planes", 2: "head chunk 0", 3: "head chunks", 4: "x' + store", 5: "LN + planes (FFN start)",
         6: "FFN iter 0", 7: "FFN iter 1", 8: "FFN loop", 9: "x'' + store", 10: "tail LN + planes", 11: "inproj chunk 0",
         12: "inproj Q", 13: "inproj K", 14: "inproj V"}
for it in range(8):
    row = st[it]
    if row[0] == 0:
        break
    prev = row[0]
    print(f"tile {it}:")
    for k in range(1, 15):
        if row[k] == 0:
            continue
        print(f"   {names[k]:28s} +{row[k] - prev:8d} cyc   (t = {row[k] - row[0]:8d})")
        prev = row[k]
    if row[32] and it == 1:      # per-step stamps of the last F chunk (15..30) and of G(nf - 2) (32..47, end 48)
        f = [int(row[k + 1] - row[k]) for k in range(15, 30)] + [int(row[32] - row[30])]
        g = [int(row[k + 1] - row[k]) for k in range(32, 48)]
        print("   steps of F(nf-1) (relu + split beside it):", f, "sum", sum(f))
        print("   steps of G(nf-2) (no side work)         :", g, "sum", sum(g))
    if it + 1 < 64 and st[it + 1][0]:
        dt, dr = st[it + 1][0] - row[0], st[it + 1][31] - row[31]
        print(f"   tile total {dt} cyc = {dr / 100.0:.1f} us  ({dt / max(dr, 1) / 10.0:.2f} GHz)")
