"""Clock stamps inside the channels_last FR backward gather (level 0, probes build): where a workgroup's time goes."""
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
from r3det.ops.feature_refine import fr_backward_nhwc, fr_backward_nhwc_index  # noqa: E402

dev = torch.device("cuda")
C, H, stride = 256, 128, 8
cl = torch.channels_last
_C.lib()
_C.set_option("frb_impl", int(os.environ.get("FRB_VARIANT", 0)))
for N in (4, 2):
    boxes = syn.fr_level_boxes(N, H, H, stride, 3, device=dev)
    sets = [tuple(torch.randn(N, C, H, H, device=dev).contiguous(memory_format=cl) for _ in range(2)) for _ in range(4)]
    ix = fr_backward_nhwc_index(boxes, N, H, H, 1.0 / stride, 1)
    for i in range(8):
        fr_backward_nhwc(sets[i % 4][0], boxes, 1.0 / stride, 1, sets[i % 4][1], overwrite=True, index=ix)
    grid = N * 512
    st = torch.zeros(grid * 8, dtype=torch.int64, device=dev)
    a = st.data_ptr()
    _C.set_option("frn_stamps_lo", ctypes.c_int32(a & 0xffffffff).value)
    _C.set_option("frn_stamps_hi", ctypes.c_int32(a >> 32).value)
    fr_backward_nhwc(sets[0][0], boxes, 1.0 / stride, 1, sets[0][1], overwrite=True, index=ix)
    torch.cuda.synchronize()
    _C.set_option("frn_stamps_lo", 0)
    _C.set_option("frn_stamps_hi", 0)
    fr_backward_nhwc(sets[0][0], boxes, 1.0 / stride, 1, sets[0][1], overwrite=True, index=ix)
    torch.cuda.synchronize()
    t = st.cpu().numpy().reshape(grid, 8).astype(np.float64)
    t -= t[:, 0].min()
    k = 100.0  # s_memrealtime: 100 MHz
    names = ["start", "rows in LDS", "barrier", "cell 0 done", "cells done"]
    print(f"N={N}: span {t[:, 4].max() / k:.1f} us over {grid} workgroups")
    for i, nm in enumerate(names):
        c = t[:, i] / k
        print(f"  {nm:12s} at   min {c.min():6.1f}  mean {c.mean():6.1f}  max {c.max():6.1f} us")
    for i in range(1, 5):
        d = (t[:, i] - t[:, i - 1]) / k
        print(f"  {names[i - 1]:>12s} -> {names[i]:12s} min {d.min():6.1f}  mean {d.mean():6.1f}  max {d.max():6.1f} us")
    st0 = np.sort(t[:, 0] / k)
    print("  tiles started by 1 / 2 / 3 / 4 / 6 / 8 us:", [int((st0 <= x).sum()) for x in (1, 2, 3, 4, 6, 8)])
    late = np.nonzero(t[:, 0] / k > 3.0)[0]
    if len(late):
        print(f"  tiles that start after 3 us: {len(late)}, block indices {late.min()} .. {late.max()}; per XCD (block & 7):",
              np.bincount(late & 7, minlength=8).tolist(), "; of blocks < 1024:", int((late < 1024).sum()))
    life = (t[:, 4] - t[:, 0]) / k
    print(f"  workgroup lifetime (thread 0): mean {life.mean():.1f} us; starts: {np.percentile(t[:, 0] / k, [0, 25, 50, 75, 100]).round(1)}")
