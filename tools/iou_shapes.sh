#!/bin/bash
# rocprofv3 kernel-trace averages of the IoU op per SURVEY 8(d) shape (v1 and v3)
set -u
R=$(pwd)
OUT=${1:-gpurun_out/iou_shapes.txt}
case $OUT in /*) ;; *) OUT=$R/$OUT ;; esac
mkdir -p $(dirname $OUT)
export TMPDIR=/tmp
cd /tmp
: > $OUT
for shp in 128x196416 512x196416 128x21824 1000x128 v3_128x196416; do
  export IOU_PROF_SHAPE=$shp
  rm -rf /tmp/kt_run
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_run -o t -- python3 $R/tools/iou_prof.py > /tmp/kt_run.log 2>&1
  echo "## $shp" >> $OUT
  grep "rbbox_iou" /tmp/kt_run.log | sed 's/^/# /' >> $OUT
  python3 $R/tools/kstats.py /tmp/kt_run iou_ fill >> $OUT
done
cat $OUT
