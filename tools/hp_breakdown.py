import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/r3det-pytorch_amd")
import torch, bench
dev = torch.device("cuda")
wl = bench.build_hot_workload(dev, 7)
from r3det.ops.feature_refine import fr_forward
from r3det.core.post_processing import multiclass_nms_rotated
from r3det.synthetic import STRIDES
def fr():
    for f, b, o, s in zip(wl["feats"], wl["boxes"], wl["outs"], STRIDES): fr_forward(f, b, 1.0 / s, 1, o)
def nms():
    for mb, ms in wl["pools"]: multiclass_nms_rotated(mb, ms, 0.05, dict(iou_thr=0.1), 2000)
for name, fn in (("fr", fr), ("nms", nms), ("both", lambda: (fr(), nms()))):
    print(name, round(bench.timeit(fn, 20, 3) * 1e3, 3), "ms")
