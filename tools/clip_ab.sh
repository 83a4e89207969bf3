#!/bin/bash
# Round 5: the straight-line v1 clip (clip_impl 0) against the LDS-list form (clip_impl 1) in the three drains --
# rocprofv3 kernel-trace averages of the IoU pipeline per shape, the NMS op at 8576 / 2000 and the fused assignment.
#   bash tools/clip_ab.sh <out.txt>      (on the GPU box, from the repo root)
set -u
R=$(pwd)
OUT=${1:-gpurun_out/clip_ab.txt}
case $OUT in /*) ;; *) OUT=$R/$OUT ;; esac
mkdir -p $(dirname $OUT)
export TMPDIR=/tmp
cd /tmp
: > $OUT
kt() { rm -rf /tmp/kt_run; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_run -o t -- "$@" > /tmp/kt_run.log 2>&1; }
for impl in 0 1; do
  for shp in 128x196416 128x21824 512x196416; do
    export IOU_PROF_SHAPE=$shp IOU_PROF_clip_impl=$impl
    kt python3 $R/tools/iou_prof.py
    echo "## clip_impl=$impl IoU v1 $shp" >> $OUT
    grep "rbbox_iou" /tmp/kt_run.log | sed 's/^/# /' >> $OUT
    python3 $R/tools/kstats.py /tmp/kt_run iou_ >> $OUT
  done
  unset IOU_PROF_SHAPE IOU_PROF_clip_impl
  for n in 8576 2000; do
    export NMS_PROF_N=$n NMS_PROF_clip_impl=$impl
    kt python3 $R/tools/nms_prof.py
    echo "## clip_impl=$impl NMS v1 n=$n" >> $OUT
    grep "batched_rnms" /tmp/kt_run.log | sed 's/^/# /' >> $OUT
    python3 $R/tools/kstats.py /tmp/kt_run nms_ mc_ >> $OUT
  done
  unset NMS_PROF_N NMS_PROF_clip_impl
  export CLIP_IMPL=$impl
  kt python3 $R/tools/assign_prof.py
  echo "## clip_impl=$impl fused assignment 128 x 196416" >> $OUT
  grep "assign" /tmp/kt_run.log | sed 's/^/# /' >> $OUT
  python3 $R/tools/kstats.py /tmp/kt_run iou_ assign_ >> $OUT
  unset CLIP_IMPL
done
cat $OUT
