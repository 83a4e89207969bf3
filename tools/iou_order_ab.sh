#!/bin/bash
# iou_stream3_kernel: which workgroups zero their tile early (option iou_order: -1 all, b = bit b of the linear index, 31 none)
#   bash tools/iou_order_ab.sh <out.txt>      (on the GPU box, from the repo root)
set -u
R=$(pwd)
OUT=${1:-gpurun_out/iou_order_ab.txt}
case $OUT in /*) ;; *) OUT=$R/$OUT ;; esac
mkdir -p $(dirname $OUT)
export TMPDIR=/tmp
cd /tmp
: > $OUT
for ord in ${ORDERS:--1 31 0 1 3 4 5 6 8 9 11}; do
  export IOU_PROF_SHAPE=128x196416 IOU_PROF_iou_order=$ord
  rm -rf /tmp/kt_run
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_run -o t -- python3 $R/tools/iou_prof.py > /tmp/kt_run.log 2>&1
  echo "## iou_order $ord" >> $OUT
  python3 $R/tools/kstats.py /tmp/kt_run iou_stream >> $OUT || tail -5 /tmp/kt_run.log >> $OUT
done
cat $OUT
