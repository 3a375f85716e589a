"""Host-side time of one forward (no device sync after it) vs the synchronised step time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superpoints_registration_amd import get_config, ops, synthetic
from superpoints_registration_amd.regtr import RegTR
dev = torch.device('cuda:0')
cfg = get_config('3dmatch')
model = RegTR(cfg); synthetic.fill_parameters(model, 0); model = model.to(dev).eval()
model.inputs_resident = True
pairs = [synthetic.make_pair(16384, seed=i) for i in range(16)]
batch = {"src_xyz": [torch.from_numpy(p[0]).to(dev) for p in pairs], "tgt_xyz": [torch.from_numpy(p[1]).to(dev) for p in pairs]}
with torch.no_grad():
    for _ in range(3):
        model(batch)
    torch.cuda.synchronize()
    host, total = [], []
    for _ in range(8):
        t0 = time.perf_counter(); model(batch); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        host.append(t1 - t0); total.append(t2 - t0)
    print('host issue time per forward (ms): %.2f   synchronised: %.2f' % (1e3 * sum(host) / len(host), 1e3 * sum(total) / len(total)))
    t0 = time.perf_counter()
    for _ in range(10):
        model(batch)
    torch.cuda.synchronize()
    print('10 back-to-back forwards: %.2f ms each' % (1e2 * (time.perf_counter() - t0)))
    # where does the host time go?  (cProfile of one forward)
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable(); model(batch); pr.disable(); torch.cuda.synchronize()
    st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(14)
