"""A/B of launch variants of the NCHW FR backward gather inside one process (option frb_impl), level 0, rotating buffers."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# launch variants that were not shipped / clock stamps live in the probes build of the library (make probes)
os.environ.setdefault("R3DET_HIP_LIB", os.path.join(ROOT, "r3det-pytorch_amd", "libr3det_hip_probes.so"))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from r3det import _C, synthetic as syn  # noqa: E402
from r3det.ops.feature_refine import fr_backward_index, fr_backward_indexed  # noqa: E402

dev = torch.device("cuda")
C, H, stride = 256, 128, 8
variants = [int(v) for v in os.environ.get("FRN_VARIANTS", "0,4").split(",")]
for N in (4, 2):
    boxes = syn.fr_level_boxes(N, H, H, stride, 3, device=dev)
    nset = max(3, int(0.9e9 // (2 * N * C * H * H * 4)))
    sets = [tuple(torch.randn(N, C, H, H, device=dev) for _ in range(2)) for _ in range(nset)]
    ix = fr_backward_index(boxes, N, C, H, H, 1.0 / stride, 1)
    for rnd in range(3):
        for var in variants:
            _C.set_option("frb_impl", var)
            for i in range(nset):
                fr_backward_indexed(sets[i][0], 1, sets[i][1], ix)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for i in range(30):
                g, o = sets[i % nset]
                fr_backward_indexed(g, 1, o, ix)
            e.record()
            torch.cuda.synchronize()
            print(f"N={N} frb_impl={var}: {s.elapsed_time(e) * 1000 / 30:6.1f} us", flush=True)
    del sets
_C.set_option("frb_impl", 0)
