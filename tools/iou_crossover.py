"""Where the stream + drain pipeline overtakes the one-launch tile kernel (option iou_impl 2 vs 4), on random boxes of
two densities."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")): sys.path.insert(0, p)
import torch
from r3det import synthetic as syn, _C
from r3det.ops import rbbox_iou
dev = torch.device("cuda")
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps
for (m, n) in [(128, 1000), (256, 1024), (500, 500), (128, 4096), (1000, 1000), (128, 8192), (2000, 512), (512, 2048), (2000, 2000), (128, 16384), (4000, 4000)]:
    a, b = syn.rand_rboxes(m, 3, device=dev), syn.rand_rboxes(n, 4, device=dev)
    row = []
    for impl in (2, 4, 0):
        _C.set_option("iou_impl", impl)
        row.append(t(lambda: rbbox_iou(a, b)))
    _C.set_option("iou_impl", 0)
    nnz = float((rbbox_iou(a, b) > 0).float().mean())
    print(f"{m:5d} x {n:6d} = {m * n / 1e6:6.2f} Mpairs  overlap {100 * nnz:5.2f} %   tile kernel {row[0]:8.1f} us   pipeline {row[1]:8.1f} us   auto {row[2]:8.1f} us", flush=True)
