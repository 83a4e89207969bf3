#!/bin/bash
# Round 6: occupancy bounds of three IoU kernels, same sources compiled twice --
#   A = the shipped library: iou_mat_compact_kernel<1, ., ., 8> / iou_vec_kernel<1> bounded to 3 waves per SIMD (168 VGPRs),
#       iou_drain3_kernel<3, fast> bounded to 3 (164 VGPRs, no scratch);
#   B = the same sources with -DR3_COMPACT_WAVES=1 -DR3_VEC_WAVES=1 -DR3_DRAIN3_V3_WAVES=4 (built here into /tmp): rounds
#       1-5's forms -- the v1 tile / aligned kernels unbounded (185 / 188 VGPRs, occupancy 2), the v3 drain at occupancy 4
#       with 12 B / lane of scratch.
# (profiles/r06_iou_occupancy_ab.txt was taken while A and B were the other way round for the first two kernels: its
# "library A" rows are the unbounded forms, "library B" the bounded ones that ship now.)
# Kernel durations under rocprofv3 --kernel-trace --stats.   bash tools/iou_occupancy_ab.sh <out.txt>
set -u
R=$(pwd)
OUT=${1:-gpurun_out/iou_occupancy_ab.txt}
case $OUT in /*) ;; *) OUT=$R/$OUT ;; esac
mkdir -p $(dirname $OUT)
export TMPDIR=/tmp
C=$R/r3det-pytorch_amd/csrc
mkdir -p /tmp/r3_iou_ab
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fvisibility=hidden \
      -fvisibility-inlines-hidden -Wno-pass-failed -DR3_COMPACT_WAVES=1 -DR3_VEC_WAVES=1 -DR3_DRAIN3_V3_WAVES=4 \
      -c $C/r3_iou.hip -o /tmp/r3_iou_ab/r3_iou.o
hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=$C/exports.map -o /tmp/r3_iou_ab/libr3det_hip_iou_ab.so \
      $C/r3_api.o /tmp/r3_iou_ab/r3_iou.o $C/r3_nms.o $C/r3_fr.o $C/r3_frb.o $C/r3_boxes.o $C/r3_pool.o $C/r3_epilogue.o $C/r3_poly.o
cd /tmp
: > $OUT
for lib in A B; do
  if [ $lib = B ]; then export R3DET_HIP_LIB=/tmp/r3_iou_ab/libr3det_hip_iou_ab.so; else unset R3DET_HIP_LIB; fi
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
