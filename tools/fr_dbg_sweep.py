"""FR forward level 0: taps from the table kernel (fr_dbg 1) vs derived in the cell kernel (fr_dbg 2)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# launch variants that were not shipped / clock stamps live in the probes build of the library (make probes)
os.environ.setdefault("R3DET_HIP_LIB", os.path.join(ROOT, "r3det-pytorch_amd", "libr3det_hip_probes.so"))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from r3det import _C, synthetic as syn  # noqa: E402
from r3det.ops.feature_refine import fr_forward  # noqa: E402

dev = torch.device("cuda")
for N in (1, 4, 16):
    feats, boxes = syn.fr_pyramid(N, 256, 9, device=dev)
    for lvl in (0, 1):
        f, b = feats[lvl], boxes[lvl]
        o = torch.empty_like(f)
        ref = torch.empty_like(f)
        _C.set_option("fr_impl", 2)
        fr_forward(f, b, 1 / syn.STRIDES[lvl], 1, ref)
        _C.set_option("fr_impl", 10)
        for dbg in (1, 2):
            _C.set_option("fr_dbg", dbg)
            for _ in range(5):
                fr_forward(f, b, 1 / syn.STRIDES[lvl], 1, o)
            assert torch.equal(o, ref)
            ts = []
            for _ in range(7):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(20):
                    fr_forward(f, b, 1 / syn.STRIDES[lvl], 1, o)
                e.record()
                torch.cuda.synchronize()
                ts.append(s.elapsed_time(e) * 1e3 / 20)
            ts.sort()
            print(f"N={N:2d} level{lvl} {'table' if dbg == 1 else 'boxes'}: med {ts[3]:7.1f} us  min {ts[0]:7.1f} us", flush=True)
_C.set_option("fr_impl", 0)
_C.set_option("fr_dbg", 0)
