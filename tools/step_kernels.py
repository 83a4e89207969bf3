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
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size") or "x".join(str(r.get(c, "")) for c in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))))
rows.sort()
tend = rows[-1][1]
sel = [r for r in rows if r[0] >= tend - int(step_ms * 1e6)]
acc = collections.defaultdict(lambda: [0, 0])
for s, e, k, _g in sel:
    acc[k[:100]][0] += 1
    acc[k[:100]][1] += e - s
tot = sum(v[1] for v in acc.values())
print(f"# kernels that started in the last {step_ms} ms of the trace: {len(sel)} launches, {tot / 1e3:.1f} us busy")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{v[0]:4d} {v[1] / 1e3:9.1f} us  {k}")
# the custom-op kernels of the WHOLE trace by grid size (one symbol serves all pyramid levels: the roofline launch
# of bench.py is the largest grid of fr_forward_nhwc*)
by = collections.defaultdict(list)
for s, e, k, g in rows:
    if "fr_forward" in k or "iou_" in k or "nms_" in k:
        by[(k[:70], g)].append((e - s) / 1e3)
print("# per symbol and grid size over the whole trace: launches, avg us, min us, max us")
for (k, g), v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print(f"{len(v):4d} {sum(v) / len(v):9.2f} {min(v):9.2f} {max(v):9.2f}  grid {g:>9s}  {k}")
