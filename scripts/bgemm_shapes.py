import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.chdir(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import runpy
from superpoints_registration_amd import autograd
autograd._BGEMM_LOG = []
sys.argv = ['scripts/train_probe.py']
runpy.run_path('scripts/train_probe.py', run_name='__main__')
c = collections.Counter(autograd._BGEMM_LOG)
for k, v in sorted(c.items(), key=lambda kv: -kv[1]):
    nb, m, n, kk, sa, sb = k
    t128 = ((m + 127) // 128) * ((n + 127) // 128)
    print(v, 'calls: batches', nb, 'm', m, 'n', n, 'k', kk, 'sa', sa, 'sb', sb, 'tiles128 x batches =', t128 * nb)
