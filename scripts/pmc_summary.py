import csv, sys, collections
f = sys.argv[1]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
seen = set()
for r in rows:
    name = r['Kernel_Name'].replace('spr::(anonymous namespace)::', '').replace('void ', '')[:40]
    agg[name][r['Counter_Name']] += float(r['Counter_Value'])
    key = (r['Dispatch_Id'])
    if key not in seen:
        seen.add(key); cnt[name] += 1
names = sorted(agg, key=lambda n: -agg[n].get('SQ_WAVE_CYCLES', agg[n].get('FETCH_SIZE', 0)))
for n in names[:14]:
    c = agg[n]; k = cnt[n]
    print(f"{n:40s} n={k:4d} " + ' '.join(f"{a}={v/k:.4g}" for a, v in sorted(c.items())))
