import numpy as np, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import native
from superpoints_registration_amd import synthetic, ops
from superpoints_registration_amd.hip_shim import cpp_neighbors
rng = np.random.default_rng(5)
a = (synthetic.box_faces(3000, rng, 1.0) + rng.normal(0, 0.004, (3000, 3))).astype(np.float32)
b = (synthetic.box_faces(2400, rng, 0.8) + rng.normal(0, 0.004, (2400, 3))).astype(np.float32)
pts, lens = np.concatenate([a, b]), np.array([3000, 2400], np.int32)
sub, sub_lens = native.ref_grid_subsample(pts, lens, 0.05, 0)
ref, mc = native.radius_neighbors(sub, pts, sub_lens, lens, 0.09, 128)
dev = torch.device('cuda')
q, s = torch.from_numpy(sub).to(dev), torch.from_numpy(pts).to(dev)
qcu, scu = ops.lengths_to_cu(sub_lens.tolist(), dev), ops.lengths_to_cu(lens.tolist(), dev)
for limit in (40, 64, 100, 128):
    for algo in (0, 1):
        out, m = ops.radius_neighbors(q, s, qcu, scu, 0.09, limit, algo=algo)
        o = out.cpu().numpy()
        w = min(limit, ref.shape[1])
        cnt_ref = (ref < len(pts)).sum(1); cnt = (o < len(pts)).sum(1)
        print('limit', limit, 'algo', algo, 'max_count', m, 'shape', o.shape, 'oracle max', mc, 'rows equal:', int((o[:, :w] == ref[:, :w]).all(1).sum()) if o.shape[1] >= w else 'narrow', '/', len(o),
              'rows with fewer valid than oracle:', int((cnt < np.minimum(cnt_ref, limit)).sum()))
nb = cpp_neighbors.batch_query(sub, pts, sub_lens, lens, radius=0.09)
print('shim', nb.shape)
