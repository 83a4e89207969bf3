#!/bin/bash
# What each part of iou_stream3_kernel adds to the plain fill of its tile (probes build; option fr_walk 1001..1003
# is read as the stream-phase probe there): 1 = zeros only, 2 = + prologue, 3 = + row loop, 0 = whole kernel;
# 16 + p = phase p without the zero stores (what the tests cost alone).
#   bash tools/iou_stream_phases.sh <out.txt>      (on the GPU box, from the repo root)
set -u
R=$(pwd)
OUT=${1:-gpurun_out/iou_stream_phases.txt}
case $OUT in /*) ;; *) OUT=$R/$OUT ;; esac
mkdir -p $(dirname $OUT)
export TMPDIR=/tmp R3DET_HIP_LIB=$R/r3det-pytorch_amd/libr3det_hip_probes.so
cd /tmp
: > $OUT
for ph in ${PHASES:-1001 1002 1003 8 1017 1018 1019 1016}; do
  export IOU_PROF_SHAPE=128x196416 IOU_PROF_fr_walk=$ph
  rm -rf /tmp/kt_run
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_run -o t -- python3 $R/tools/iou_prof.py > /tmp/kt_run.log 2>&1
  echo "## phase $ph" >> $OUT
  python3 $R/tools/kstats.py /tmp/kt_run iou_stream >> $OUT || tail -5 /tmp/kt_run.log >> $OUT
done
cat $OUT
