"""Samples GPU power / shader clock (sysfs hwmon, falling back to rocm-smi) while a workload
loops: is the 'phases add' behaviour a power-cap effect (clock drops under combined load)?
usage: python scripts/power_probe.py [attn|attn2|bench|idle] [seconds]"""
import glob, json, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def find_hwmon():
    out = []
    for h in glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*'):
        files = {os.path.basename(f) for f in glob.glob(h + '/*')}
        out.append((h, sorted(f for f in files if f.startswith(('power', 'freq')))))
    return out

def read(p):
    try:
        return open(p).read().strip()
    except Exception as e:
        return None

samples = []
stop = False
def sampler(hws_):
    while not stop:
        row = {'t': time.time()}
        for i, hw in enumerate(hws_):
            for k in ('power1_input', 'freq1_input'):
                v = read(hw + '/' + k)
                if v is not None:
                    row['%s.%d' % (k, i)] = int(v)
        samples.append(row)
        time.sleep(0.02)

def smi_sampler():
    while not stop:
        try:
            o = subprocess.run(['rocm-smi', '--showpower', '--showclocks', '--json'], capture_output=True, text=True, timeout=10).stdout
            samples.append({'t': time.time(), 'smi': o[:1500]})
        except Exception as e:
            samples.append({'t': time.time(), 'err': str(e)})
        time.sleep(0.3)

what = sys.argv[1] if len(sys.argv) > 1 else 'attn'
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
hws = find_hwmon()
print('hwmon count:', len(hws))
use = [h for h, f in hws if any(x.startswith('power1') for x in f)]
import torch
from superpoints_registration_amd import ops
dev = torch.device('cuda:0')
th = threading.Thread(target=sampler, args=(use,)) if use else threading.Thread(target=smi_sampler)
th.start()
t_end = time.time() + secs
n = 0
if what.startswith('attn'):
    nseg, L, nhead = 32, 1930, 8
    T = nseg * L
    g = torch.Generator(device='cpu'); g.manual_seed(0)
    qkv = torch.randn(T, 768, generator=g).to(dev)
    cu = (torch.arange(0, nseg + 1, dtype=torch.int32) * L).to(dev)
    kv = (torch.arange(nseg, dtype=torch.int32) ^ 1).to(dev)
    out = torch.empty(T, 256, device=dev)
    ops.set_attn_mode(2 if what == 'attn2' else (0 if what == 'attn0' else 1))
    q, k, v = qkv[:, :256], qkv[:, 256:512], qkv[:, 512:]
    while time.time() < t_end:
        for _ in range(50):
            ops.attention(q, k, v, cu, kv, L, nhead, out=out)
        torch.cuda.synchronize(); n += 50
elif what == 'gemm':
    x = torch.randn(61745, 256, device=dev); w = torch.randn(1024, 256, device=dev) * 0.05; b = torch.zeros(1024, device=dev)
    while time.time() < t_end:
        for _ in range(50):
            ops.linear(x, w, b, act=1)
        torch.cuda.synchronize(); n += 50
elif what == 'copy':
    a = torch.empty(1 << 28, device=dev); b = torch.empty(1 << 28, device=dev)
    while time.time() < t_end:
        for _ in range(20):
            b.copy_(a)
        torch.cuda.synchronize(); n += 20
else:
    time.sleep(secs)
dt = secs
stop = True; th.join()
print(what, 'iterations', n, 'us/iter %.1f' % (dt / max(n, 1) * 1e6))
keys = sorted({k for s in samples for k in s if k != 't' and k not in ('smi', 'err')})
for k in keys:
    v = [s[k] for s in samples if k in s]
    if max(v) - min(v) > 0.2 * max(v) or what == 'idle':
        print(k, 'n', len(v), 'min', min(v), 'median', sorted(v)[len(v) // 2], 'max', max(v), 'tail', v[-5:])
for s in samples[:: max(1, len(samples) // 6)]:
    if 'smi' in s or 'err' in s:
        print(s)
