"""Per-queue busy time and the largest gaps of the main queue from a rocprofv3 --kernel-trace CSV of bench.py:
how much a captured (HIP graph) forward could still recover.  python scripts/stream_gaps.py <dir> [window_ms]"""
import collections, csv, glob, sys

d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/final/stats"
win = float(sys.argv[2]) if len(sys.argv) > 2 else 250.0
import os
f = max(glob.glob(d + "/*/*kernel_trace.csv"), key=os.path.getmtime)          # gpurun merges without deleting: newest
rows = list(csv.DictReader(open(f)))
loop = [r for r in rows if "k_xenc_chain" in r["Kernel_Name"] or "k_attn_h3" in r["Kernel_Name"]]
tmax = max(int(r["End_Timestamp"]) for r in loop)            # end of the last timed forward's transformer
t0 = tmax - int(win * 1e6)
byq = collections.defaultdict(list)
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s >= t0 and e <= tmax:
        byq[r["Queue_Id"]].append((s, e, r["Kernel_Name"]))
print(f"{f}\nwindow: the last {win:.0f} ms before the end of the last transformer kernel of the timed loop")
main = max(byq, key=lambda q: sum(e - s for s, e, _ in byq[q]))
for q, l in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    l.sort()
    busy, (cs, ce) = 0, l[0][:2]
    for s, e, _ in l[1:]:
        if s > ce:
            busy += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    busy += ce - cs
    span = l[-1][1] - l[0][0]
    print(f"queue {q}{' (main)' if q == main else ''}: {len(l)} kernels, busy {busy / 1e6:.1f} ms of {span / 1e6:.1f} ms = {100 * busy / span:.1f} %")
l = sorted(byq[main])
gaps = collections.defaultdict(lambda: [0, 0])
tot = 0
def short(n):
    return n.replace("spr::(anonymous namespace)::", "").replace("void ", "")[:44]
for (s0, e0, n0), (s1, e1, n1) in zip(l, l[1:]):
    g = s1 - e0
    if g > 0:
        tot += g
        gaps[(short(n0), short(n1))][0] += g
        gaps[(short(n0), short(n1))][1] += 1
print(f"main queue: {tot / 1e6:.2f} ms of gaps in the window; the largest by pair of neighbours:")
for (a, b), (g, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:12]:
    print(f"  {g / 1e6:6.2f} ms  {c:4d} x {g / c / 1e3:7.1f} us   {a}  ->  {b}")
