"""One training step (fwd + loss + bwd + clip + AdamW) at BASELINE size: per-stage milliseconds."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from superpoints_registration_amd import get_config, synthetic
from superpoints_registration_amd.regtr import RegTR
from superpoints_registration_amd.training import Trainer

dev = torch.device('cuda:0')
B = int(os.environ.get('PAIRS', 4)); N = int(os.environ.get('POINTS', 16384))
cfg = get_config('3dmatch')
model = RegTR(cfg); synthetic.fill_parameters(model, 0); model = model.to(dev)
pairs = [synthetic.make_pair(N, seed=i) for i in range(B)]
rng = np.random.default_rng(777)
batch = {"src_xyz": [torch.from_numpy(p[0]).to(dev) for p in pairs], "tgt_xyz": [torch.from_numpy(p[1]).to(dev) for p in pairs],
         "pose": torch.from_numpy(np.stack([p[2] for p in pairs]).astype(np.float32)).to(dev),
         "src_overlap": [torch.from_numpy(rng.random(len(p[0])) < 0.6).to(dev) for p in pairs],
         "tgt_overlap": [torch.from_numpy(rng.random(len(p[1])) < 0.6).to(dev) for p in pairs]}
tr = Trainer(cfg).setup(model)
ev = lambda: torch.cuda.Event(enable_timing=True)
for step in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e = [ev() for _ in range(5)]
    model.train()
    e[0].record()
    pred = model(dict(batch, **{}))
    b2 = dict(batch); b2['kpconv_meta'] = None
    e[1].record()
    # (the forward wrote kpconv_meta into the dict it was given)
    torch.cuda.synchronize()
    print('forward ok', flush=True) if step == 0 else None
    break
# full steps through the Trainer, timing stages with events inside a copy of its logic
for step in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e = [ev() for _ in range(5)]
    model.train()
    bb = dict(batch)
    e[0].record()
    pred = model(bb)
    e[1].record()
    losses = model.compute_loss(pred, bb)
    tr.optimizer.zero_grad()
    e[2].record()
    losses['total'].backward()
    e[3].record()
    torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=tr.grad_clip)
    tr.optimizer.step(); tr.scheduler.step()
    e[4].record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('step %d: total %.1f ms (%.1f pairs/s) | forward %.1f | loss %.1f | backward %.1f | clip+AdamW %.1f | loss %.4f | peak mem %.1f GB' % (
        step, dt * 1e3, B / dt, e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2]), e[2].elapsed_time(e[3]), e[3].elapsed_time(e[4]),
        float(losses['total']), torch.cuda.max_memory_allocated() / 1e9), flush=True)
