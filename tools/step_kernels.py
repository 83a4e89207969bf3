"""Per-kernel busy time of the LAST step in a rocprofv3 --kernel-trace csv directory.
    python tools/step_kernels.py <dir> <step_ms> [top]"""
import collections
import csv
import glob
import sys

d, step_ms = sys.argv[1], float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
tend = rows[-1][1]
sel = [r for r in rows if r[0] >= tend - int(step_ms * 1e6)]
acc = collections.defaultdict(lambda: [0, 0])
for s, e, k in sel:
    acc[k[:100]][0] += 1
    acc[k[:100]][1] += e - s
tot = sum(v[1] for v in acc.values())
print(f"# kernels that started in the last {step_ms} ms of the trace: {len(sel)} launches, {tot / 1e3:.1f} us busy")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{v[0]:4d} {v[1] / 1e3:9.1f} us  {k}")
