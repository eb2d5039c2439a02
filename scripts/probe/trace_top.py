"""top kernels of a rocprofv3 --kernel-trace --stats run: python3 scripts/probe/trace_top.py <dir> [n]"""
import csv, glob, os, sys
d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
files = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
if not files:
    print("no kernel_stats.csv under", d); sys.exit(0)
rows = list(csv.DictReader(open(files[0])))
for r in rows[:n]:
    print("%-110s calls %6s  avg_us %9.1f  %5s%%" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
