"""Clock stamps inside iou_drain3_kernel<1> (probes build): where a drain workgroup's time goes at 128 x 196 416.
CLIP_IMPL=0|1 (straight-line / LDS-list clip), IOU_DWGS (workgroups), IOU_SHAPE (128x196416 | 128x21824 | 512x196416)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("R3DET_HIP_LIB", os.path.join(ROOT, "r3det-pytorch_amd", "libr3det_hip_probes.so"))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from r3det import _C, synthetic as syn  # noqa: E402
from r3det.ops import rbbox_iou  # noqa: E402

dev = torch.device("cuda")
_C.lib()
impl = int(os.environ.get("CLIP_IMPL", 0))
_C.set_option("clip_impl", impl)
dw = int(os.environ.get("IOU_DWGS", 0))
_C.set_option("iou_dwgs", dw)
shape = os.environ.get("IOU_SHAPE", "128x196416")
anchors = syn.anchor_grid(device=dev)
gt = syn.dota_like_rboxes(512 if shape.startswith("512") else 128, 6 if shape.startswith("512") else 5, device=dev)
cols = anchors if shape.endswith("196416") else torch.cat(
    [syn.fr_level_boxes(1, 1024 // s, 1024 // s, s, 50 + i, device=dev) for i, s in enumerate(syn.STRIDES)])
for _ in range(5):
    rbbox_iou(gt, cols)
grid = dw if dw > 0 else 1536
st = torch.zeros(grid * 8, dtype=torch.int64, device=dev)
a = st.data_ptr()
_C.set_option("frn_stamps_lo", ctypes.c_int32(a & 0xffffffff).value)
_C.set_option("frn_stamps_hi", ctypes.c_int32(a >> 32).value)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
out = rbbox_iou(gt, cols)
e.record()
torch.cuda.synchronize()
_C.set_option("frn_stamps_lo", 0)
_C.set_option("frn_stamps_hi", 0)
t = st.cpu().numpy().reshape(grid, 8).astype(np.float64)
redo = (t[:, 7].astype(np.int64) >> 16).astype(np.float64)
trips = (t[:, 7].astype(np.int64) & 0xffff).astype(np.float64)
used = t[:, 0] > 0
t = t[used]
trips = trips[used]
redo = redo[used]
t0 = t[:, 0].min()
k = 100.0  # s_memrealtime: 100 MHz
print(f"clip_impl {impl} shape {shape} workgroups {used.sum()} of {grid}: call {s.elapsed_time(e) * 1e3:.1f} us (with stamps); "
      f"stamp span {(t[:, 6].max() - t0) / k:.1f} us; trips of wave 0: min {trips.min():.0f} mean {trips.mean():.2f} max {trips.max():.0f}; "
      f"nnz {int((out > 0).sum())}")
names = ["start", "prefix", "entry", "records", "clipped", "stored", "end"]
for i, nm in enumerate(names):
    c = (t[:, i] - t0) / k
    c = c[t[:, i] > 0]
    if c.size:
        print(f"  {nm:8s} at   min {c.min():6.1f}  mean {c.mean():6.1f}  max {c.max():6.1f} us   ({c.size} workgroups)")
for i in range(1, 7):
    m = (t[:, i] > 0) & (t[:, i - 1] > 0)
    d = (t[m, i] - t[m, i - 1]) / k
    if d.size:
        print(f"  {names[i - 1]:>8s} -> {names[i]:8s} min {d.min():6.1f}  mean {d.mean():6.1f}  max {d.max():6.1f} us")

end = (t[:, 6] - t0) / k
for tr in sorted(set(trips.tolist())):
    for rd in sorted(set(redo.tolist())):
        m = (trips == tr) & (redo == rd)
        if m.any():
            print(f"  wave 0 with {tr:.0f} trips, {rd:.0f} redo passes: {m.sum():5d} workgroups, end min {end[m].min():6.1f} mean {end[m].mean():6.1f} max {end[m].max():6.1f} us")
print("  end-time deciles:", np.round(np.percentile(end, [10, 25, 50, 75, 90, 95, 99, 100]), 1))
