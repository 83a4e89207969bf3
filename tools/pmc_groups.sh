#!/bin/bash
# PMC passes (one counter group per run, --kernel-trace only, as the pool requires) for a tool script; prints the
# per-kernel averages of every counter for kernels whose name contains <filter>.
# usage (GPU box, repo root): bash tools/pmc_groups.sh <outfile> <kernel-name-filter> "<group1>;<group2>;..." <script.py> [args]
set -u
OUT=$1; FILT=$2; PMCG=$3; shift 3
R=$(pwd)
export TMPDIR=/tmp
cd /tmp
: > $R/$OUT
IFS=";" read -ra GS <<< "$PMCG"
i=0
for grp in "${GS[@]}"; do
  i=$((i+1))
  rm -rf /tmp/pmc_g_$i
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc_g_$i -o p -- python3 $R/"$@" > /tmp/pmc_g_$i.log 2>&1
  f=$(find /tmp/pmc_g_$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    python3 - "$f" "$grp" "$FILT" >> $R/$OUT <<'PY'
import csv, sys, collections
f, grp, filt = sys.argv[1], sys.argv[2], sys.argv[3]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(f)):
    k = row.get("Kernel_Name", "")
    if filt not in k:
        continue
    acc[(k[:100], row.get("Grid_Size", ""))][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("## group:", grp)
for (k, g), d in sorted(acc.items()):
    print(k, "grid", g, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
  else
    echo "## group: $grp -> no csv" >> $R/$OUT; tail -3 /tmp/pmc_g_$i.log >> $R/$OUT
  fi
done
cat $R/$OUT
