#!/bin/bash
# PMC passes for any tool script (one counter group per run, --kernel-trace only).
# usage (on the GPU box, from the repo root):  bash tools/pmc_run.sh <outfile> <kernel-name-filter> <script.py> [args]
set -u
OUT=$1; FILT=$2; shift 2
R=$(pwd)
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rm -rf /tmp/pmc_run_$i
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc_run_$i -o p -- python3 $R/"$@" > /tmp/pmc_run_$i.log 2>&1
  f=$(find /tmp/pmc_run_$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    python3 - "$f" "$grp" "$FILT" >> $R/$OUT <<'PY'
import csv, sys, collections
f, grp, filt = sys.argv[1], sys.argv[2], sys.argv[3]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(f)):
    k = row.get("Kernel_Name", "")
    if filt not in k: continue
    g = row.get("Grid_Size", "")
    acc[(k[:70], g)][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("## group:", grp)
for (k, g), d in sorted(acc.items()):
    print(k, "grid", g, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
  else
    echo "## group: $grp -> no csv"; tail -3 /tmp/pmc_run_$i.log
  fi
done
cat $R/$OUT
