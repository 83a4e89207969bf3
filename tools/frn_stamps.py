"""Clock stamps inside frn_gather_kernel (level 0): where a workgroup's time goes.  FR_BWD_N (4)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# launch variants that were not shipped / clock stamps live in the probes build of the library (make probes)
os.environ.setdefault("R3DET_HIP_LIB", os.path.join(ROOT, "r3det-pytorch_amd", "libr3det_hip_probes.so"))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from r3det import _C, synthetic as syn  # noqa: E402
from r3det.ops.feature_refine import fr_backward_index, fr_backward_indexed  # noqa: E402

dev = torch.device("cuda")
C, H, stride = 256, 128, 8
var = int(os.environ.get("FRN_VARIANT", 0))
_C.lib()
_C.set_option("frb_impl", var)
for N in (int(os.environ.get("FR_BWD_N", 4)), 2):
    boxes = syn.fr_level_boxes(N, H, H, stride, 3, device=dev)
    if os.environ.get('FR_BWD_FIELD') == 'none':
        boxes[:, 0] = -1000.0   # every sample out of range: no entries at all
    nset = 4
    sets = [tuple(torch.randn(N, C, H, H, device=dev) for _ in range(2)) for _ in range(nset)]
    ix = fr_backward_index(boxes, N, C, H, H, 1.0 / stride, 1)
    for i in range(8):
        fr_backward_indexed(sets[i % nset][0], 1, sets[i % nset][1], ix)
    grid = N * C // 2
    st = torch.zeros(grid * 8, dtype=torch.int64, device=dev)
    a = st.data_ptr()
    _C.set_option("frn_stamps_lo", ctypes.c_int32(a & 0xffffffff).value)
    _C.set_option("frn_stamps_hi", ctypes.c_int32(a >> 32).value)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    fr_backward_indexed(sets[0][0], 1, sets[0][1], ix)
    e.record()
    torch.cuda.synchronize()
    _C.set_option("frn_stamps_lo", 0)
    _C.set_option("frn_stamps_hi", 0)
    t = st.cpu().numpy().reshape(grid, 8).astype(np.float64)
    t -= t[:, 0].min()
    us = s.elapsed_time(e) * 1000
    k = 100.0   # s_memrealtime: 100 MHz
    print(f"variant {var} N={N}: event {us:.1f} us (with stamps); stamp span {t[:, 5].max() / k:.1f} us")
    names = ["start", "staged", "barrier", "gathered", "drained", "end"]
    for i, nm in enumerate(names):
        c = t[:, i] / k
        print(f"  {nm:9s} at   min {c.min():6.1f}  mean {c.mean():6.1f}  max {c.max():6.1f} us")
    for i in range(1, 6):
        d = (t[:, i] - t[:, i - 1]) / k
        print(f"  {names[i - 1]:>9s} -> {names[i]:9s} min {d.min():6.1f}  mean {d.mean():6.1f}  max {d.max():6.1f} us")
    first = t[:, 0] < np.median(t[:, 0])
    for nm, m in (("first-round workgroups", first), ("second-round workgroups", ~first)):
        d = (t[m][:, 1:6] - t[m][:, 0:5]) / k
        print(f"  {nm}: start at {t[m][:, 0].mean() / k:5.1f}; phases (mean us):", np.round(d.mean(0), 1))
