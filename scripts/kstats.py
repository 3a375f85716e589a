"""Prints the top rows of a rocprofv3 kernel_stats.csv found under a directory: name, calls, average us, total ms."""
import csv, glob, sys
d = sys.argv[1]; top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
div = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
fs = sorted(glob.glob(d + "/**/*kernel_stats.csv", recursive=True), key=lambda f: -sum(1 for _ in open(f)))
rows = list(csv.DictReader(open(fs[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total %.3f ms (/%g = %.3f)" % (tot / 1e6, div, tot / 1e6 / div))
for r in rows[:top]:
    print("%-110s %6s %9.1f us %8.3f ms" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6 / div))
