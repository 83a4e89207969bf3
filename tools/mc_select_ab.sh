#!/bin/bash
# candidate selection of the batched multiclass pipeline: one launch (round 5) against count + write (option nms_impl 5)
#   bash tools/mc_select_ab.sh <out.txt>
set -u
R=$(pwd)
OUT=${1:-gpurun_out/mc_select_ab.txt}
case $OUT in /*) ;; *) OUT=$R/$OUT ;; esac
mkdir -p $(dirname $OUT)
export TMPDIR=/tmp
cd /tmp
: > $OUT
for impl in 0 5; do
  export NMS_PROF_nms_impl=$impl
  rm -rf /tmp/kt_run
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_run -o t -- python3 $R/tools/nms_prof.py > /tmp/kt_run.log 2>&1
  echo "## nms_impl $impl" >> $OUT
  grep "multiclass" /tmp/kt_run.log | sed 's/^/# /' >> $OUT
  python3 $R/tools/kstats.py /tmp/kt_run mc_ >> $OUT
done
cat $OUT
