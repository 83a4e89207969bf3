"""FR backward cell kernel: what do the LDS float atomics cost?  fr_dbg 11 = plain stores at the
same addresses, 12 = atomics at conflict-free addresses, 13 = one atomic per position."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402

from r3det import _C, synthetic as syn  # noqa: E402
from r3det.ops.feature_refine import fr_backward  # noqa: E402

feats, boxes = syn.fr_pyramid(4, 256, 9, device="cuda")
f, b = feats[0], boxes[0]
g = torch.empty_like(f)
for dbg in (0, 100, 11, 12, 13, 0):  # 0: packed path (default); 100: the atomic cell kernel
    _C.set_option("fr_dbg", 0 if dbg == 100 else dbg)
    _C.set_option("fr_impl", 10 if dbg == 100 else 0)
    for _ in range(3):
        fr_backward(f, b, 1.0 / 8, 1, g, overwrite=True)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        fr_backward(f, b, 1.0 / 8, 1, g, overwrite=True)
    e.record()
    torch.cuda.synchronize()
    print(f"fr_dbg {dbg:2d}: {s.elapsed_time(e) * 1e3 / 20:8.1f} us", flush=True)
_C.set_option("fr_dbg", 0)
_C.set_option("fr_impl", 0)
f1, b1 = feats[1], boxes[1]
g1 = torch.empty_like(f1)
for impl in (0, 10):
    _C.set_option("fr_impl", impl)
    for _ in range(3):
        fr_backward(f1, b1, 1.0 / 16, 1, g1, overwrite=True)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        fr_backward(f1, b1, 1.0 / 16, 1, g1, overwrite=True)
    e.record()
    torch.cuda.synchronize()
    print(f"level 1 fr_impl {impl:2d}: {s.elapsed_time(e) * 1e3 / 20:8.1f} us", flush=True)
_C.set_option("fr_impl", 0)
