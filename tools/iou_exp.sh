#!/bin/bash
# experiment: kernel trace of the IoU op at 128x196416 for a list of iou_qcap values (0 = default)
#   bash tools/iou_exp.sh "0 2"
set -u
R=$(pwd)
export TMPDIR=/tmp IOU_PROF_SHAPE=${IOU_PROF_SHAPE:-128x196416}
cd /tmp
for d in $1; do
  export IOU_PROF_${IOU_EXP_OPT:-iou_qcap}=$d
  rm -rf /tmp/kt_run
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_run -o t -- python3 $R/tools/iou_prof.py > /tmp/kt_run.log 2>&1
  echo "## ${IOU_EXP_OPT:-iou_qcap}=$d"; grep "rbbox_iou" /tmp/kt_run.log
  python3 $R/tools/kstats.py /tmp/kt_run iou_ fill
done
