"""Where the wall time of bench.py's hot path goes: host time to enqueue each part (no synchronisation in between)
against the GPU time of the whole step."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402
from r3det.core.post_processing import multiclass_nms_rotated_batch  # noqa: E402
from r3det.ops import fr_boxes  # noqa: E402
from r3det.ops.feature_refine import fr_module_levels_nhwc  # noqa: E402

dev = torch.device("cuda")
wl = bench.build_hot_workload(dev, seed=7)
L = wl["levels"]


def fr():
    fr_module_levels_nhwc([lv["a"] for lv in L], [lv["b"] for lv in L], wl["bias"], wl["bias"], [lv["res"] for lv in L],
                          [lv["boxes"] for lv in L], [lv["scale"] for lv in L], 1, [lv["out"] for lv in L])


def pool():
    fr_boxes.levels_pool([lv["cls"] for lv in L], [lv["reg"] for lv in L], [lv["rois"] for lv in L], 1, 15, 2000,
                         (bench.IMG, bench.IMG), wl["pool_boxes"], wl["pool_scores"])


def nms():
    return multiclass_nms_rotated_batch(wl["pool_boxes"], wl["pool_scores"], bench.SCORE_THR, bench.NMS_CFG,
                                        bench.MAX_PER_IMG, hint=wl["nms_hint"])


for _ in range(5):
    fr(); pool(); nms()
torch.cuda.synchronize()
acc = {"fr": 0.0, "pool": 0.0, "nms (incl. its final read)": 0.0}
n = 50
t0 = time.perf_counter()
for _ in range(n):
    t = time.perf_counter(); fr(); acc["fr"] += time.perf_counter() - t
    t = time.perf_counter(); pool(); acc["pool"] += time.perf_counter() - t
    t = time.perf_counter(); nms(); acc["nms (incl. its final read)"] += time.perf_counter() - t
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) / n
for k, v in acc.items():
    print(f"host time in {k:28s}: {v / n * 1e6:7.1f} us")
print(f"step wall: {tot * 1e6:7.1f} us")
for name, fn in (("fr", fr), ("pool", pool), ("nms", nms)):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        fn()
    e.record()
    torch.cuda.synchronize()
    print(f"{name} alone, back to back: {s.elapsed_time(e) * 1000 / 20:7.1f} us per call")
