#!/bin/bash
# the drain's blocks of 64 entries by atomic ticket (iou_dyn 1) against the static stride (0): IoU v1 / v3 and the assignment
#   bash tools/iou_dyn_ab.sh <out.txt>
set -u
R=$(pwd)
OUT=${1:-gpurun_out/iou_dyn_ab.txt}
case $OUT in /*) ;; *) OUT=$R/$OUT ;; esac
mkdir -p $(dirname $OUT)
export TMPDIR=/tmp
cd /tmp
: > $OUT
for dyn in 0 1 0 1; do
  for shp in 128x196416 512x196416 128x21824 v3_128x196416; do
    export IOU_PROF_SHAPE=$shp IOU_PROF_iou_dyn=$dyn
    rm -rf /tmp/kt_run
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_run -o t -- python3 $R/tools/iou_prof.py > /tmp/kt_run.log 2>&1
    echo "## iou_dyn $dyn $shp" >> $OUT
    python3 $R/tools/kstats.py /tmp/kt_run iou_drain >> $OUT || tail -5 /tmp/kt_run.log >> $OUT
  done
done
cat $OUT
