#!/bin/bash
# HBM traffic of one kernel from separate PMC passes (FETCH_SIZE, WRITE_SIZE), per launch.
# usage (GPU box, repo root): bash tools/pmc_traffic.sh <outfile> <kernel-name-filter> <script.py> [args]
set -u
OUT=$1; FILT=$2; shift 2
R=$(pwd)
export TMPDIR=/tmp
cd /tmp
: > $R/$OUT
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pmc_tr
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc_tr -o p -- python3 $R/"$@" > /tmp/pmc_tr.log 2>&1
  f=$(find /tmp/pmc_tr -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$grp" "$FILT" >> $R/$OUT <<'PY'
import csv, sys, collections
f, grp, filt = sys.argv[1], sys.argv[2], sys.argv[3]
acc = collections.defaultdict(list)
for row in csv.DictReader(open(f)):
    k = row.get("Kernel_Name", "")
    if filt in k and row["Counter_Name"] == grp:
        acc[k[:110]].append(float(row["Counter_Value"]))
for k, v in acc.items():
    print(grp, round(sum(v) / len(v), 1), "n=", len(v), k)
PY
done
cat $R/$OUT
