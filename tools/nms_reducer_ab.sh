#!/bin/bash
# The batched pipeline's reducer, rounds (nms_impl 0) against the walk in score order (nms_impl 4): rocprofv3 averages of
# the reducer kernel on the survey's pools (tools/nms_prof.py) and on the bench model's own pool.
set -u
R=$(pwd)
OUT=${1:-gpurun_out/nms_reducer_ab.txt}
case $OUT in /*) ;; *) OUT=$R/$OUT ;; esac
mkdir -p $(dirname $OUT)
export TMPDIR=/tmp
cd /tmp
: > $OUT
for impl in 0 4; do
  for n in 2000 5344 8576; do
    rm -rf /tmp/kt_run
    NMS_PROF_N=$n NMS_PROF_nms_impl=$impl rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_run -o t -- python3 $R/tools/nms_prof.py > /tmp/kt.log 2>&1
    echo "## nms_impl=$impl batched_rnms n=$n" >> $OUT
    grep batched /tmp/kt.log | sed 's/^/# /' >> $OUT
    python3 $R/tools/kstats.py /tmp/kt_run nms_reduce >> $OUT
  done
  rm -rf /tmp/kt_run
  R3DET_NMS_IMPL=$impl rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_run -o t -- python3 $R/bench.py --steps 10 --warmup 4 --model-only > /tmp/kt.log 2>&1
  echo "## nms_impl=$impl bench model's own pool (B = 4, 15 classes in equal shares)" >> $OUT
  python3 $R/tools/kstats.py /tmp/kt_run nms_ mc_ >> $OUT
done
cat $OUT
