"""The module-tail launch (level 0, N = 4, rotating buffers) against how far the boxes sample from their own
(transposed) cell: centre = cell + N(0, sigma cells), sigma = 0.4 (bench.py's field) ... 4, the "trained" field of
tools/fr_bwd_prof.py (every 4 x 4 block of positions regresses to one centre) and the bench model's own stage-1 boxes
when FR_MODEL_FIELD=1.  Both forms of the kernel (option fr_dbg 0 auto = wide / 9 pairs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# launch variants that were not shipped / clock stamps live in the probes build of the library (make probes)
os.environ.setdefault("R3DET_HIP_LIB", os.path.join(ROOT, "r3det-pytorch_amd", "libr3det_hip_probes.so"))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")): sys.path.insert(0, p)
import torch
from r3det import synthetic as syn, _C
from r3det.ops.feature_refine import fr_module_nhwc
dev = torch.device("cuda")
N, C, H, stride = 4, 256, 128, 8
cl = torch.channels_last
sets = [tuple(torch.randn(N, C, H, H, device=dev).contiguous(memory_format=cl) for _ in range(4)) for _ in range(3)]
ba, bb = torch.randn(C, device=dev), torch.randn(C, device=dev)
fields = []
for sigma in (0.4, 1.0, 2.0, 4.0):
    fields.append((f"sigma {sigma:3.1f} cells", syn.fr_level_boxes(N, H, H, stride, 3, jitter=sigma / 4, device=dev)))
tb = syn.fr_level_boxes(N, H, H, stride, 3, device=dev)
g = (tb[:, :2] / (4 * stride)).floor() * (4 * stride) + 2 * stride
tb[:, :2] = g + torch.randn_like(g) * 0.3 * stride
fields.append(("trained (4 x 4 piles)", tb))
for name, bx in fields:
    res = []
    for dbg in (0, 9):
        _C.set_option("fr_dbg", dbg)
        for i in range(6):
            a, b, r, o = sets[i % 3]; fr_module_nhwc(a, b, ba, bb, r, bx, 1 / stride, 1, o)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(30):
            a, b, r, o = sets[i % 3]; fr_module_nhwc(a, b, ba, bb, r, bx, 1 / stride, 1, o)
        e.record(); torch.cuda.synchronize()
        res.append(s.elapsed_time(e) * 1000 / 30)
    print(f"{name:24s}: wide {res[0]:6.1f} us   pairs {res[1]:6.1f} us", flush=True)
_C.set_option("fr_dbg", 0)
