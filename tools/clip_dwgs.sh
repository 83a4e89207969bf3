#!/bin/bash
# drain workgroup-count sweep for the straight-line clip (clip_impl 0): IoU 128 x 196 416 and the fused assignment
set -u
R=$(pwd)
OUT=$R/gpurun_out/clip_dwgs.txt
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd /tmp
: > $OUT
kt() { rm -rf /tmp/kt_run; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_run -o t -- "$@" > /tmp/kt_run.log 2>&1; }
for d in ${DWGS:-512 768 1024 1280 1536 2048}; do
  export IOU_PROF_SHAPE=128x196416 IOU_PROF_iou_dwgs=$d
  kt python3 $R/tools/iou_prof.py
  echo "## iou_dwgs=$d IoU v1 128x196416" >> $OUT
  grep "rbbox_iou" /tmp/kt_run.log | sed 's/^/# /' >> $OUT
  python3 $R/tools/kstats.py /tmp/kt_run iou_drain >> $OUT
  export IOU_PROF_SHAPE=128x21824
  kt python3 $R/tools/iou_prof.py
  echo "## iou_dwgs=$d IoU v1 128x21824" >> $OUT
  python3 $R/tools/kstats.py /tmp/kt_run iou_drain >> $OUT
  unset IOU_PROF_SHAPE IOU_PROF_iou_dwgs
  export IOU_DWGS=$d
  kt python3 $R/tools/assign_prof.py
  echo "## iou_dwgs=$d assignment" >> $OUT
  grep "^assign" /tmp/kt_run.log | sed 's/^/# /' >> $OUT
  python3 $R/tools/kstats.py /tmp/kt_run assign_drain iou_drain >> $OUT
  unset IOU_DWGS
done
cat $OUT
