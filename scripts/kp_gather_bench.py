"""Pure-gather ceiling of the KPConv neighbour gather on the bench pyramid (VERDICT r2 task 1a).

Builds the 16-pair bench pyramid with the product preprocessor, then gathers the rows the three
KPConv levels read (128 / 256 / 512-byte feature rows) with scripts/abl/kp_gather.hip -- no influence
math, no MFMA -- and prints GB/s per CU next to the guide's table.  Output: gpurun_out/kp_gather.txt
"""
import ctypes, os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superpoints_registration_amd import get_config, synthetic
from superpoints_registration_amd.regtr import RegTR

here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, 'abl', 'libkp_gather.so')
if not os.path.exists(so):
    subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '--offload-arch=gfx950', '-shared', '-fPIC',
                           os.path.join(here, 'abl', 'kp_gather.hip'), '-o', so])
L = ctypes.CDLL(so)
vp, i_ = ctypes.c_void_p, ctypes.c_int
L.kpg_run.argtypes = [i_, i_, i_, i_, i_, vp, i_, i_, i_, vp, i_, i_, i_, ctypes.POINTER(ctypes.c_float),
                      ctypes.POINTER(ctypes.c_ulonglong)]
L.kpg_run.restype = i_

dev = torch.device('cuda:0')
B = 16
cfg = get_config('3dmatch')
model = RegTR(cfg); synthetic.fill_parameters(model, 0); model = model.to(dev).eval()
pairs = [synthetic.make_pair(16384, seed=i) for i in range(B)]
meta = model.preprocessor([torch.from_numpy(p[0]).to(dev) for p in pairs] + [torch.from_numpy(p[1]).to(dev) for p in pairs])
os.makedirs(os.path.join(os.path.dirname(here), 'gpurun_out'), exist_ok=True)
out = open(os.path.join(os.path.dirname(here), 'gpurun_out', 'kp_gather.txt'), 'w')


def say(*a):
    s = ' '.join(str(v) for v in a)
    print(s, flush=True); out.write(s + '\n'); out.flush()


def morton_perm(pts, lens, cell):
    """query order: cloud by cloud, 3-D Morton code of floor((p - min) / cell)"""
    cloud = torch.repeat_interleave(torch.arange(len(lens), device=pts.device), lens.long())
    mn = torch.zeros((len(lens), 3), device=pts.device).index_reduce_(0, cloud, pts, 'amin', include_self=False)
    c = ((pts - mn[cloud]) / cell).floor().long().clamp_(0, 1023)
    code = torch.zeros(pts.shape[0], dtype=torch.long, device=pts.device)
    for b in range(10):
        for a in range(3):
            code |= ((c[:, a] >> b) & 1) << (3 * b + a)
    key = cloud * (1 << 30) + code
    return torch.argsort(key, stable=True)


def run(tag, nb, ns, x, row_bytes, mode, depth, wpw, wpc, xcd, reps=5):
    ms = ctypes.c_float(0); rows = ctypes.c_ulonglong(0)
    rc = L.kpg_run(mode, depth, row_bytes, wpw, wpc, nb.data_ptr(), nb.shape[0], nb.stride(0), nb.shape[1],
                   x.data_ptr(), ns, xcd, reps, ctypes.byref(ms), ctypes.byref(rows))
    if rc != 0:
        say(tag, 'rc', rc); return
    gb = rows.value * row_bytes / 1e9
    say('%-24s mode %d D %2d waves/CU %2d xcd %d : %7.1f us  %5.2f GB  %6.2f TB/s  %5.1f GB/s/CU  (in flight/CU %3d KiB)' % (
        tag, mode, depth, wpw * wpc, xcd, ms.value * 1e3, gb, gb / ms.value, gb / ms.value * 1e3 / 256, depth * wpw * wpc))


for lvl, ch in ((1, 64), (0, 32), (2, 128)):
    pts = meta['points'][lvl]; nb = meta['_i32'][('neighbors', lvl)].contiguous()
    lens = meta['stack_lengths'][lvl].to(dev)
    n = pts.shape[0]
    x = torch.rand((n, ch), device=dev)
    rb = ch * 4
    valid = int((nb < n).sum())
    say('== level %d: %d points, nbr %s, valid %d (%.1f per row), table %.1f MB, %d-byte rows' % (
        lvl, n, tuple(nb.shape), valid, valid / n, n * rb / 1e6, rb))
    radius = cfg.first_subsampling_dl * cfg.conv_radius * (2 ** lvl)
    perm = morton_perm(pts, lens, radius)
    nb_m = nb[perm].contiguous()
    rnd = torch.randint(0, n, nb.shape, device=dev, dtype=torch.int32)
    rnd = torch.where(nb < n, rnd, torch.full_like(nb, n))
    seq = (torch.arange(n, device=dev, dtype=torch.int32)[:, None] + torch.arange(nb.shape[1], device=dev, dtype=torch.int32)[None]).clamp(max=n - 1)
    seq = torch.where(nb < n, seq, torch.full_like(nb, n))
    # main sweep on the real matrix in reference order
    for mode in (0, 1, 2):
        for depth, wpw, wpc in ((8, 8, 1), (16, 8, 1), (4, 8, 1), (2, 8, 1), (4, 16, 1), (8, 12, 1), (16, 4, 1), (16, 4, 2), (8, 4, 4), (4, 4, 1), (8, 4, 1)):
            if mode != 0 and depth > 8 and wpw * wpc > 8:
                continue
            run('real/hash-order', nb, n, x, rb, mode, depth, wpw, wpc, 0)
    for tag, m, xcd in (('real/morton', nb_m, 0), ('real/morton+xcd-chunk', nb_m, 1), ('uniform-random', rnd, 0),
                        ('sequential rows', seq, 0)):
        for mode in (0, 1):
            for depth, wpw, wpc in ((8, 8, 1), (16, 8, 1), (4, 8, 1)):
                run(tag, m, n, x, rb, mode, depth, wpw, wpc, xcd)
out.close()
