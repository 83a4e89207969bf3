#!/bin/bash
# mc_sort_prepare_kernel under rocprofv3: batched_rnms at n = 8576 and the hot path's 4-image call (tools/nms_timeline.py)
export TMPDIR=/tmp
R=$(pwd)
cd /tmp
export NMS_PROF_N=8576
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_a -o t -- python3 $R/tools/nms_prof.py > /tmp/log_a 2>&1
python3 $R/tools/kstats.py /tmp/kt_a mc_sort nms_stream | grep -v calls
unset NMS_PROF_N
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o t -- python3 $R/tools/nms_timeline.py run > /tmp/log_b 2>&1
python3 $R/tools/nms_timeline.py show /tmp/tl | grep "mc_sort\|nms_stream"
