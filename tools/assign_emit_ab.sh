#!/bin/bash
# What the fused assignment's drain pays for what it emits (probes build): 0 as shipped, 1 no column keys, 2 a look
# at the key before the column atomic (the form until round 5), 3 no siou store, 4 no row keys, 5 no row flush, 6 a look before
# the flush's atomics (until round 5).   bash tools/assign_emit_ab.sh <out.txt>
set -u
R=$(pwd)
OUT=${1:-gpurun_out/assign_emit_ab.txt}
case $OUT in /*) ;; *) OUT=$R/$OUT ;; esac
mkdir -p $(dirname $OUT)
export TMPDIR=/tmp R3DET_HIP_LIB=$R/r3det-pytorch_amd/libr3det_hip_probes.so
cd /tmp
: > $OUT
for pb in ${PROBES:-0 1 2 3 4 5 6}; do
  export ASSIGN_PROBE=$pb
  rm -rf /tmp/kt_run
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_run -o t -- python3 $R/tools/assign_prof.py > /tmp/kt_run.log 2>&1
  echo "## probe $pb" >> $OUT
  grep "^assign" /tmp/kt_run.log | sed 's/^/# /' >> $OUT
  python3 $R/tools/kstats.py /tmp/kt_run iou_ assign_ zero >> $OUT || tail -5 /tmp/kt_run.log >> $OUT
done
cat $OUT
