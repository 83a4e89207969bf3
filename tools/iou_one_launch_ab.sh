#!/bin/bash
# Round 6: kernel durations (rocprofv3 --kernel-trace --stats) of the IoU matrix op in the shipped form (iou_impl 0:
# stream = tests + zeros, then drain) and in the one-launch form (iou_impl 5: K1 = tests + survivor bits, K2 = drain whose
# first workgroups write the zeros), per shape; then the same-process A/B of tools/iou_one_launch_ab.py (per call, outputs
# compared bit for bit).
#   bash tools/iou_one_launch_ab.sh <out.txt>      (on the GPU box, from the repo root)
set -u
R=$(pwd)
OUT=${1:-gpurun_out/iou_one_launch_ab.txt}
case $OUT in /*) ;; *) OUT=$R/$OUT ;; esac
mkdir -p $(dirname $OUT)
export TMPDIR=/tmp
cd /tmp
: > $OUT
for shp in ${SHAPES:-128x196416 512x196416 128x21824}; do
  for impl in 0 5; do
    export IOU_PROF_SHAPE=$shp IOU_PROF_iou_impl=$impl
    rm -rf /tmp/kt_run
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_run -o t -- python3 $R/tools/iou_prof.py > /tmp/kt_run.log 2>&1
    echo "## $shp iou_impl $impl" >> $OUT
    grep "rbbox_iou" /tmp/kt_run.log | sed 's/^/# /' >> $OUT
    python3 $R/tools/kstats.py /tmp/kt_run iou_ >> $OUT || tail -5 /tmp/kt_run.log >> $OUT
  done
done
unset IOU_PROF_SHAPE IOU_PROF_iou_impl
echo "## same process, alternating (per call, no profiler)" >> $OUT
python3 $R/tools/iou_one_launch_ab.py 2>&1 | grep -v amdgpu.ids >> $OUT
cat $OUT
