"""Per kernel symbol AND grid size of a rocprofv3 --kernel-trace csv directory: launches, avg / min / max us (one
symbol serves all pyramid levels and both models: the grid tells them apart).
    python tools/kgrid.py <dir> [substr ...]"""
import collections
import csv
import glob
import sys

d, subs = sys.argv[1], sys.argv[2:]
by = collections.defaultdict(list)
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if subs and not any(x in k for x in subs):
            continue
        g = r.get("Grid_Size") or "x".join(str(r.get(c, "")) for c in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
        by[(k[:90], g)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(f"{'calls':>6} {'avg_us':>9} {'min_us':>9} {'max_us':>9}  grid / kernel")
for (k, g), v in sorted(by.items(), key=lambda kv: (kv[0][0], -sum(kv[1]))):
    print(f"{len(v):6d} {sum(v) / len(v):9.2f} {min(v):9.2f} {max(v):9.2f}  grid {g:>14s}  {k}")
