#!/bin/bash
# Round 6: occupancy bounds of three IoU kernels, same sources compiled twice --
#   A = the shipped library: iou_mat_compact_kernel<1,..> / iou_vec_kernel<1> unbounded (185 / 188 VGPRs, occupancy 2),
#       iou_drain3_kernel<3, fast> bounded to 3 waves per SIMD (164 VGPRs, no scratch);
#   B = tools/scratch/ab/libr3det_hip_iou_ab.so (-DR3_COMPACT_WAVES=3 -DR3_VEC_WAVES=3 -DR3_DRAIN3_V3_WAVES=4): 168 VGPRs at
#       occupancy 3 for the first two, the v3 drain at occupancy 4 with 12 B / lane of scratch (round 5's form).
# Kernel durations under rocprofv3 --kernel-trace --stats.   bash tools/iou_occupancy_ab.sh <out.txt>
set -u
R=$(pwd)
OUT=${1:-gpurun_out/iou_occupancy_ab.txt}
case $OUT in /*) ;; *) OUT=$R/$OUT ;; esac
mkdir -p $(dirname $OUT)
export TMPDIR=/tmp
cd /tmp
: > $OUT
for lib in A B; do
  if [ $lib = B ]; then export R3DET_HIP_LIB=$R/tools/scratch/ab/libr3det_hip_iou_ab.so; else unset R3DET_HIP_LIB; fi
  for shp in 1000x128 v3_128x196416 vec; do
    export IOU_PROF_SHAPE=$shp
    rm -rf /tmp/kt_run
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_run -o t -- python3 $R/tools/iou_prof.py > /tmp/kt_run.log 2>&1
    echo "## library $lib, $shp" >> $OUT
    grep "rbbox_iou" /tmp/kt_run.log | sed 's/^/# /' >> $OUT
    python3 $R/tools/kstats.py /tmp/kt_run iou_ >> $OUT || tail -5 /tmp/kt_run.log >> $OUT
  done
done
cat $OUT
