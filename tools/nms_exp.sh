#!/bin/bash
# experiment: kernel trace of batched_rnms for a list of pool sizes
#   bash tools/nms_exp.sh "2000 5344 8576"
set -u
R=$(pwd)
export TMPDIR=/tmp
cd /tmp
for n in $1; do
  export NMS_PROF_N=$n
  rm -rf /tmp/kt_run
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_run -o t -- python3 $R/tools/nms_prof.py > /tmp/kt_run.log 2>&1
  grep "batched_rnms" /tmp/kt_run.log
  python3 $R/tools/kstats.py /tmp/kt_run nms_ mc_ fill
done
