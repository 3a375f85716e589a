"""Turns gpurun_out/final/ (scripts/measure_round.sh) into the committed profiles/ files.

  profiles/<tag>_bench.json                 bench.py's JSON line (with cpu_baseline)
  profiles/<tag>_bench_under_rocprof.json   the same command under rocprofv3
  profiles/<tag>_kernel_stats.csv           rocprofv3 --kernel-trace --stats summary
  profiles/<tag>_pmc_hbm_summary.txt        FETCH_SIZE / WRITE_SIZE per kernel (separate passes)
  profiles/kpconv_traffic.json              HBM bytes per launch of the roofline kernel (read by bench.py)

HBM bytes follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are reported in KiB-like
units of 1024 B; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, so reads are doubled.
"""
import collections, csv, datetime, glob, json, os, shutil, sys


def newest(pattern):
    """gpurun merges into gpurun_out/ without deleting earlier runs: take the latest file."""
    return max(glob.glob(pattern), key=os.path.getmtime)

tag = sys.argv[1] if len(sys.argv) > 1 else "r02_final"
src = "gpurun_out/final"
os.makedirs("profiles", exist_ok=True)
shutil.copy(f"{src}/bench.json", f"profiles/{tag}_bench.json")
shutil.copy(f"{src}/bench_under_rocprof.json", f"profiles/{tag}_bench_under_rocprof.json")
stats = newest(f"{src}/stats/*/*kernel_stats.csv")
shutil.copy(stats, f"profiles/{tag}_kernel_stats.csv")
if glob.glob(f"{src}/stats_attn2/*/*kernel_stats.csv"):
    shutil.copy(newest(f"{src}/stats_attn2/*/*kernel_stats.csv"), f"profiles/{tag}_attn_mode2_kernel_stats.csv")
    shutil.copy(f"{src}/bench_attn2_under_rocprof.json", f"profiles/{tag}_attn_mode2_bench_under_rocprof.json")
if glob.glob(f"{src}/stats_attn3/*/*kernel_stats.csv"):
    shutil.copy(newest(f"{src}/stats_attn3/*/*kernel_stats.csv"), f"profiles/{tag}_attn_mode3_kernel_stats.csv")
    shutil.copy(f"{src}/bench_attn3_under_rocprof.json", f"profiles/{tag}_attn_mode3_bench_under_rocprof.json")


def per_kernel(path, counter):
    agg, n = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        agg[r["Kernel_Name"]] += float(r["Counter_Value"])
        n[r["Kernel_Name"]].add(r["Dispatch_Id"])
    return {k: (agg[k] / len(n[k]), len(n[k])) for k in agg}


fetch = per_kernel(newest(f"{src}/pmc_fetch/*/*counter_collection.csv"), "FETCH_SIZE")
write = per_kernel(newest(f"{src}/pmc_write/*/*counter_collection.csv"), "WRITE_SIZE")
bench = json.load(open(f"{src}/bench.json"))
roof = bench["roofline"]
lines = ["kernel | launches | FETCH_SIZE (KiB/launch, raw) | WRITE_SIZE (KiB/launch) | HBM bytes/launch = 2*FETCH+WRITE"]
best = None
for k in sorted(fetch, key=lambda k: -fetch[k][0] * fetch[k][1]):
    f, c = fetch[k]
    w = write.get(k, (0.0, 0))[0]
    hbm = (2.0 * f + w) * 1024.0
    lines.append(f"{k[:110]} | {c} | {f:.1f} | {w:.1f} | {hbm:.4g}")
open(f"profiles/{tag}_pmc_hbm_summary.txt", "w").write("\n".join(lines) + "\n")

# roofline kernel: match the instantiation bench.py reported (cin -> CC template argument)
cin, cout = int(roof["code"]) // 100000, int(roof["code"]) % 100000
# the ring kernel's mangled name carries <CC = cin, COUT, NS> (k_kpconv_ringILi64ELi64ELi4EE...); shapes served by
# the streamed kernel (k_kpconv_mfma<CC, TQ, ...>) are matched on the average launch time instead
stats_rows = {r["Name"]: r for r in csv.DictReader(open(stats))}
ring = [k for k in fetch if f"k_kpconv_ringILi{cin}ELi{cout}E" in k]
if ring:
    pick = ring[0]
else:
    cands = {k: v for k, v in fetch.items() if "k_kpconv_mfma" in k}
    target_ns = roof["avg_launch_ms"] * 1e6
    pick = min(cands, key=lambda k: abs(float(stats_rows[k]["AverageNs"]) - target_ns) if k in stats_rows else 1e30)
f, c = fetch[pick]
w = write.get(pick, (0.0, 0))[0]
out = dict(tag=tag, collected_utc=datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%M:%SZ"),
           code=cin * 100000 + cout, kernel_avg_ms=roof["avg_launch_ms"],
           kernel=pick, hbm_bytes_per_launch=int((2.0 * f + w) * 1024.0), fetch_size_raw_kib=f, write_size_kib=w,
           launches_sampled=c, rocprof_avg_ns=float(stats_rows[pick]["AverageNs"]),
           bench_avg_launch_ms=roof["avg_launch_ms"], cin=cin, cout=cout,
           note="HBM bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024, gfx950 FETCH_SIZE correction per MI355X_MICROARCH.md")
json.dump(out, open("profiles/kpconv_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
print("alg bytes/launch", roof["alg_bytes_per_launch"])
