"""Experiment: one 16-pair step as two concurrent 8-pair forwards on two HIP streams (two host threads)."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superpoints_registration_amd import get_config, ops, synthetic
from superpoints_registration_amd.regtr import RegTR
dev = torch.device('cuda:0')
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cfg = get_config('3dmatch')
model = RegTR(cfg); synthetic.fill_parameters(model, 0); model = model.to(dev).eval()
pairs = [synthetic.make_pair(16384, seed=i) for i in range(B)]
groups = [pairs[i::NS] for i in range(NS)]
batches = [{"src_xyz": [torch.from_numpy(p[0]).to(dev) for p in g], "tgt_xyz": [torch.from_numpy(p[1]).to(dev) for p in g]} for g in groups]
streams = [torch.cuda.Stream(device=dev) for _ in range(NS)]

def work(i):
    with torch.cuda.stream(streams[i]), torch.no_grad():
        model(batches[i])

def step():
    if NS == 1:
        work(0)
    else:
        ths = [threading.Thread(target=work, args=(i,)) for i in range(NS)]
        for t in ths: t.start()
        for t in ths: t.join()

for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 8
for _ in range(K): step()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"streams={NS}: {B * K / dt:.1f} pairs/s, {dt / K * 1e3:.2f} ms/step")
