"""Print the per-kernel averages of a rocprofv3 --kernel-trace --stats csv directory (optionally only kernels
whose name contains one of the given substrings).   python tools/kstats.py <dir> [substr ...]"""
import csv
import glob
import sys

d, subs = sys.argv[1], sys.argv[2:]
for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
    print(f"{'calls':>6} {'avg_us':>9} {'min_us':>9} {'max_us':>9} {'pct':>6}  kernel")
    for r in csv.DictReader(open(f)):
        if subs and not any(x in r["Name"] for x in subs):
            continue
        print(f"{int(r['Calls']):6d} {float(r['AverageNs']) / 1e3:9.2f} {float(r['MinNs']) / 1e3:9.2f} "
              f"{float(r['MaxNs']) / 1e3:9.2f} {float(r['Percentage']):6.2f}  {r['Name'][:110]}")
