import csv, sys, glob
f = sys.argv[1]
nfwd = float(sys.argv[2]) if len(sys.argv) > 2 else 7
rows = list(csv.DictReader(open(f)))
tot = sum(int(r['TotalDurationNs']) for r in rows)
print(f'total kernel ms {tot/1e6:.2f}; per forward {tot/1e6/nfwd:.2f} ms')
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 22]:
    n = r['Name'].replace('spr::(anonymous namespace)::', '').replace('void ', '')
    print(f"{n[:58]:58s} calls {int(r['Calls']):5d} tot {int(r['TotalDurationNs'])/1e6:8.2f} ms avg {float(r['AverageNs'])/1e3:8.1f} us {float(r['Percentage']):5.1f}%")
