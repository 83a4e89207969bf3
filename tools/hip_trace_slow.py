"""List the slowest HIP API calls of a rocprofv3 --hip-trace csv (which call held the 86 ms stall?).
    python tools/hip_trace_slow.py <dir with *_hip_api_trace.csv> [top]"""
import csv
import glob
import sys

d = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
rows = []
for f in glob.glob(d + "/**/*hip_api_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["Start_Timestamp"]), r["Function"]))
if not rows:
    print("no hip api trace found under", d)
    sys.exit(0)
t_end = max(r[1] for r in rows)
rows.sort(reverse=True)
print(f"{len(rows)} HIP API calls; slowest {top} (duration ms, seconds before the end of the trace, function)")
for dur, st, fn in rows[:top]:
    print(f"{dur / 1e6:10.3f} ms   t_end-{(t_end - st) / 1e9:8.3f} s   {fn}")
