"""Summarise a rocprofv3 --kernel-trace CSV per (kernel, grid): count, median, min microseconds.
    python tools/kt_by_grid.py <kernel_trace.csv> [name-filter ...]"""
import collections
import csv
import re
import sys

acc = collections.defaultdict(list)
filters = sys.argv[2:]
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if filters and not any(f in k for f in filters):
        continue
    short = re.sub(r"\(anonymous namespace\)::", "", k)
    short = re.sub(r"\(.*", "", short)[-44:]
    grid = "x".join(r.get(c, "?") for c in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
    acc[(short, grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (k, g), v in sorted(acc.items()):
    v.sort()
    print(f"{k:46s} grid {g:18s} n={len(v):4d}  med {v[len(v) // 2]:9.1f} us  min {v[0]:9.1f} us")
