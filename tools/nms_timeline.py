"""Timeline of one batched multiclass NMS call of the bench's hot path (kernel start / end offsets from a rocprofv3
--kernel-trace csv):  rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python3 tools/nms_timeline.py run
then  python3 tools/nms_timeline.py show DIR"""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
    sys.path.insert(0, p)

if sys.argv[1] == "run":
    import torch
    import bench
    from r3det.core.post_processing import multiclass_nms_rotated_batch
    from r3det.ops import fr_boxes
    dev = torch.device("cuda")
    wl = bench.build_hot_workload(dev, seed=7)
    L = wl["levels"]
    fr_boxes.levels_pool([lv["cls"] for lv in L], [lv["reg"] for lv in L], [lv["rois"] for lv in L], 1, 15, 2000,
                         (bench.IMG, bench.IMG), wl["pool_boxes"], wl["pool_scores"])
    for _ in range(12):
        multiclass_nms_rotated_batch(wl["pool_boxes"], wl["pool_scores"], bench.SCORE_THR, bench.NMS_CFG,
                                     bench.MAX_PER_IMG, hint=wl["nms_hint"])
    torch.cuda.synchronize()
else:
    f = glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    ks = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
    # the last call: from the last mc_count kernel on
    idx = [i for i, k in enumerate(ks) if "mc_count" in k[0]]
    a, b = idx[-2], idx[-1]
    t0 = ks[a][1]
    for name, s, e in ks[a:b]:
        print(f"{(s - t0) / 1e3:8.1f} .. {(e - t0) / 1e3:8.1f} us  ({(e - s) / 1e3:6.1f})  {name[:70]}")
    print(f"next call starts at {(ks[b][1] - t0) / 1e3:8.1f} us")
