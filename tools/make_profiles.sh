#!/bin/bash
# Round profiles (run on the GPU box from the repo root through gpurun): rocprofv3 --kernel-trace --stats summaries
# of the bench step, the training step, configs[1], the IoU and NMS ops, and the PMC passes (separate runs,
# --kernel-trace only) of the roofline kernel and of the IoU / NMS kernels.  Output: gpurun_out/profiles_<tag>/,
# copied into profiles/ by hand (tracked).
#   bash tools/make_profiles.sh r06
set -u
TAG=${1:-r06}
R=$(pwd)
O=$R/gpurun_out/profiles_$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
kt() {  # kt <outfile> <header> <program args...>: kernel trace + stats of one command
  local out=$1 hdr=$2; shift 2
  rm -rf /tmp/kt_run
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_run -o t -- "$@" > /tmp/kt_run.log 2>&1
  { echo "# rocprofv3 --kernel-trace --stats -- $hdr"; grep -v "^W2\|^E2\|^I2" /tmp/kt_run.log | tail -18 | sed 's/^/# /'; } > $out
}
# 1. bench step (R3Det inference, configs[2])
kt $O/${TAG}_bench_kernel_stats.txt "python3 bench.py --steps 10 --warmup 4 --model-only" python3 $R/bench.py --steps 10 --warmup 4 --model-only
grep '^{' /tmp/kt_run.log | tail -1 > $O/${TAG}_bench_under_rocprof.json
MS=$(python3 -c "import json;print(json.load(open('$O/${TAG}_bench_under_rocprof.json'))['ms_per_step'])")
python3 $R/tools/step_kernels.py /tmp/kt_run $MS 60 >> $O/${TAG}_bench_kernel_stats.txt
# 2. training step (configs[4])
kt $O/${TAG}_train_kernel_stats.txt "python3 tools/train_prof.py (last step)" python3 $R/tools/train_prof.py
MS=$(grep "train step" /tmp/kt_run.log | tail -1 | sed 's/.*: \([0-9.]*\) ms/\1/')
python3 $R/tools/step_kernels.py /tmp/kt_run $MS 45 >> $O/${TAG}_train_kernel_stats.txt
# 3. RRetinaNet step (configs[1])
kt $O/${TAG}_rretinanet_kernel_stats.txt "python3 tools/rretina_prof.py (last step)" python3 $R/tools/rretina_prof.py
python3 $R/tools/step_kernels.py /tmp/kt_run 10.7 40 >> $O/${TAG}_rretinanet_kernel_stats.txt
# 4. IoU op, per shape
: > $O/${TAG}_iou_kernel_stats.txt
for shp in 128x196416 512x196416 128x21824 1000x128 v3_128x196416; do
  export IOU_PROF_SHAPE=$shp
  kt /tmp/kt_one.txt "python3 tools/iou_prof.py  (IOU_PROF_SHAPE=$shp)" python3 $R/tools/iou_prof.py
  cat /tmp/kt_one.txt >> $O/${TAG}_iou_kernel_stats.txt
  python3 $R/tools/kstats.py /tmp/kt_run iou_ fill >> $O/${TAG}_iou_kernel_stats.txt
done
unset IOU_PROF_SHAPE
# 4b. plain against prepared columns (r3det_iou_prepare_columns), same process
python3 $R/tools/iou_prepared_ab.py > $O/${TAG}_iou_prepared_ab.txt 2>&1
# 5. NMS op, per size, and the batched pipeline
: > $O/${TAG}_nms_kernel_stats.txt
for n in 2000 5344 8576 16384 32768 v3_8576; do
  export NMS_PROF_N=$n
  kt /tmp/kt_one.txt "python3 tools/nms_prof.py  (NMS_PROF_N=$n)" python3 $R/tools/nms_prof.py
  cat /tmp/kt_one.txt >> $O/${TAG}_nms_kernel_stats.txt
  python3 $R/tools/kstats.py /tmp/kt_run nms_ mc_ fill >> $O/${TAG}_nms_kernel_stats.txt
done
unset NMS_PROF_N
# 5b. round 6, large pools: the sorted-chunk form (nms_impl 6) against the counting form (7) per size, no profiler; the
#     suppressor counts of those pools; the reducer's and the ranking kernel's phase stamps at 32 768 rows (probe build)
cd $R
{ for n in 8576 10240 12288 14336 16384 20000 24576 32768; do for impl in 7 6; do echo -n "nms_impl $impl: "; NMS_PROF_nms_impl=$impl NMS_PROF_N=$n python3 tools/nms_prof.py 2>&1 | grep batched; done; done; } > $O/${TAG}_nms_sorted_chunks_ab.txt
python3 tools/nms_suppressor_stats.py 8576 16384 32768 2>&1 | grep "^n=" > $O/${TAG}_nms_suppressors.txt
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I r3det-pytorch_amd/csrc -I include -o /tmp/nrp tools/probes/nms_reduce_probe.hip > /dev/null 2>&1
DUMP_POOL=/tmp/pool32k.bin NMS_PROF_N=32768 python3 tools/nms_prof.py > /dev/null 2>&1
{ echo "# tools/probes/nms_reduce_probe 32768 <the pool of tools/nms_prof.py> <label group>: s_memtime stamps of one reducer workgroup and of workgroup 0 of the ranking kernel"; for g in 0 5 9; do echo "# label group $g"; /tmp/nrp 32768 /tmp/pool32k.bin $g 2>&1 | tail -8; done; } > $O/${TAG}_nms_reduce_stamps.txt
cd /tmp
# 6. the roofline kernel alone (rotating buffers) + its PMC traffic passes
kt $O/${TAG}_fr_nhwc_kernel_stats.txt "python3 tools/fr_nhwc_prof.py" python3 $R/tools/fr_nhwc_prof.py
python3 $R/tools/kstats.py /tmp/kt_run fr_forward >> $O/${TAG}_fr_nhwc_kernel_stats.txt
cd $R
bash tools/pmc_groups.sh gpurun_out/profiles_$TAG/${TAG}_fr_nhwc_pmc.txt fr_forward_nhwc "FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum;SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM;SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" tools/fr_nhwc_prof.py > /dev/null
python3 tools/make_roofline_pmc.py gpurun_out/profiles_$TAG/${TAG}_fr_nhwc_pmc.txt gpurun_out/profiles_$TAG/roofline_kernel_pmc.json
# (bench.py below reads the tracked copy: this run's own counters, for this r3_fr.hip)
cp gpurun_out/profiles_$TAG/roofline_kernel_pmc.json profiles/roofline_kernel_pmc.json
# 6b. FR backward, both layouts (rotating buffers): level 0 / 1 at N = 4, level 0 at N = 2, and the "trained" field
: > $O/${TAG}_fr_backward_kernel_stats.txt
for cfg in "0 4 regular" "1 4 regular" "0 2 regular" "0 4 trained" "0 4 adversarial"; do
  set -- $cfg
  export FR_BWD_LEVEL=$1 FR_BWD_N=$2 FR_BWD_FIELD=$3
  kt /tmp/kt_one.txt "python3 tools/fr_bwd_prof.py  (FR_BWD_LEVEL=$1 FR_BWD_N=$2 FR_BWD_FIELD=$3)" python3 $R/tools/fr_bwd_prof.py
  cat /tmp/kt_one.txt >> $O/${TAG}_fr_backward_kernel_stats.txt
  python3 $R/tools/kstats.py /tmp/kt_run frb_ frn_ fr_b >> $O/${TAG}_fr_backward_kernel_stats.txt
done
unset FR_BWD_LEVEL FR_BWD_N FR_BWD_FIELD
cd $R
bash tools/pmc_groups.sh gpurun_out/profiles_$TAG/${TAG}_fr_backward_pmc.txt fr "FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum;SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM;SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" tools/fr_bwd_prof.py > /dev/null
cd /tmp
# 6b'. the training step's FR forward + backward over the pyramid (one autograd node for the five levels) and the
#      assignment, kernels by name
kt $O/${TAG}_train_hot_path_kernel_stats.txt "python3 tools/train_hot_path.py" python3 $R/tools/train_hot_path.py
grep "ms" /tmp/kt_run.log | tail -8 | sed 's/^/# /' >> $O/${TAG}_train_hot_path_kernel_stats.txt
python3 $R/tools/kstats.py /tmp/kt_run fr_ frb_ frn_ iou_ assign_ >> $O/${TAG}_train_hot_path_kernel_stats.txt
# 6c. the plain samplers (the reference's API: r3det_feature_refine_forward NCHW, _forward_nhwc) per level, N = 4
kt $O/${TAG}_fr_forward_kernel_stats.txt "python3 tools/fr_fwd_prof.py" python3 $R/tools/fr_fwd_prof.py
python3 $R/tools/kstats.py /tmp/kt_run fr_ >> $O/${TAG}_fr_forward_kernel_stats.txt
# 6c'. the NCHW pyramid pass: coarse levels one grid against one launch per level (both in one process)
kt $O/${TAG}_fr_levels_ab.txt "python3 tools/fr_levels_ab.py" python3 $R/tools/fr_levels_ab.py
python3 $R/tools/kstats.py /tmp/kt_run fr_forward frn_ frb_ fr_cell >> $O/${TAG}_fr_levels_ab.txt
python3 $R/tools/fr_levels_ab.py 2>/dev/null | grep "^N=" | sed 's/^/# (no profiler) /' >> $O/${TAG}_fr_levels_ab.txt
# 6c''. points = 5, NCHW: the points kernel against the plane kernel, per level (no profiler)
python3 $R/tools/fr_p5_ab.py > $O/${TAG}_fr_p5_ab.txt 2>/dev/null
python3 $R/tools/fr_p5_nhwc_ab.py >> $O/${TAG}_fr_p5_ab.txt 2>/dev/null
# 6d. the pre-NMS pool at the two models' shapes: per level (r3det_level_pool) and the whole head in one call (r3det_levels_pool)
kt $O/${TAG}_pool_kernel_stats.txt "python3 tools/pool_prof.py" python3 $R/tools/pool_prof.py
python3 $R/tools/kt_by_grid.py $(find /tmp/kt_run -name "*kernel_trace.csv" | head -1) pool_ fill >> $O/${TAG}_pool_kernel_stats.txt
# 7. PMC of the IoU and NMS kernels
cd $R
IOU_PROF_SHAPE=128x196416 bash tools/pmc_groups.sh gpurun_out/profiles_$TAG/${TAG}_iou_pmc.txt iou_ "FETCH_SIZE;WRITE_SIZE;SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR;SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" tools/iou_prof.py > /dev/null
NMS_PROF_N=8576 bash tools/pmc_groups.sh gpurun_out/profiles_$TAG/${TAG}_nms_pmc.txt nms_ "FETCH_SIZE;WRITE_SIZE;SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR;SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" tools/nms_prof.py > /dev/null
ls -la $O
# 7b. round 5: the straight-line v1 clip against the LDS-list form in the three drains (same process per form), the
#     drain's phase stamps (probes build), the whole-step graph against the eager step
bash tools/clip_ab.sh gpurun_out/profiles_$TAG/${TAG}_clip_ab.txt > /dev/null 2>&1
{ for impl in 0 1; do CLIP_IMPL=$impl IOU_DWGS=$((impl == 0 ? 1024 : 1536)) python3 tools/iou_drain_stamps.py 2>&1 | grep -v amdgpu.ids; done; } > $O/${TAG}_iou_drain_stamps.txt
python3 tools/whole_step_probe.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_whole_step_graph.txt
python3 tools/pool_edge_stats.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_model_pool_edges.txt
POOL_SPREAD=0 python3 tools/pool_edge_stats.py 2>&1 | grep -v amdgpu.ids | tail -1 | sed 's/^/one-label calibration: /' >> $O/${TAG}_model_pool_edges.txt
# 7c. round 5, second half: what each part of the IoU stream kernel adds to the plain fill (probes build), where its zeros
#     are issued, what the assignment's drain pays for what it emits (probes build), the candidate selection in one launch
bash tools/iou_stream_phases.sh gpurun_out/profiles_$TAG/${TAG}_iou_stream_phases.txt > /dev/null 2>&1
ORDERS="-1 31 0 3 8" bash tools/iou_order_ab.sh gpurun_out/profiles_$TAG/${TAG}_iou_order_ab.txt > /dev/null 2>&1
bash tools/assign_emit_ab.sh gpurun_out/profiles_$TAG/${TAG}_assign_emit_ab.txt > /dev/null 2>&1
bash tools/mc_select_ab.sh gpurun_out/profiles_$TAG/${TAG}_mc_select_ab.txt > /dev/null 2>&1
bash tools/iou_dyn_ab.sh gpurun_out/profiles_$TAG/${TAG}_iou_dyn_ab.txt > /dev/null 2>&1
# 7d. round 6: the IoU matrix with fill and clip in one launch against the shipped pair (kernel stats per form + the
#     same-process A/B), the FR node's host time against torch's own floor for the call pattern (both layouts)
bash tools/iou_one_launch_ab.sh gpurun_out/profiles_$TAG/${TAG}_iou_one_launch_ab.txt > /dev/null 2>&1
{ python3 tools/fr_host_prof.py 2>&1 | grep -E "host|wall"; echo "# channels_last:"; CL=1 python3 tools/fr_host_prof.py 2>&1 | grep -E "host|wall"; } | grep -v "fr_host_prof.py" > $O/${TAG}_fr_host_time.txt
# 8. the bench records themselves (no profiler attached): <tag>_bench_line.json = the compact stdout line of the contract,
#    <tag>_bench.json = the detail record bench.py writes next to itself (bench_detail.json)
python3 $R/bench.py --steps 30 --warmup 5 > $O/${TAG}_bench_line.json 2> /dev/null
cp $R/bench_detail.json $O/${TAG}_bench.json
python3 $R/bench.py --mode train --steps 10 --warmup 3 > /dev/null 2>&1
cp $R/bench_detail.json $O/${TAG}_train.json
python3 $R/bench.py --mode rretinanet --steps 20 --warmup 5 > /dev/null 2>&1
cp $R/bench_detail.json $O/${TAG}_rretinanet.json
