"""A/B of the channels_last module-tail kernel's launch variants (option fr_dbg: 0 shipped = the wide regions form |
9 the 4 x 4 tile pairs | 6 pairs without the non-temporal interior identity rows | 3 round-2 form), level 0 / 1,
rotating buffers; every variant must be bit-identical to the first."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# launch variants that were not shipped / clock stamps live in the probes build of the library (make probes)
os.environ.setdefault("R3DET_HIP_LIB", os.path.join(ROOT, "r3det-pytorch_amd", "libr3det_hip_probes.so"))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")): sys.path.insert(0, p)
import torch
from r3det import synthetic as syn, _C
from r3det.ops.feature_refine import fr_module_nhwc
dev = torch.device("cuda")
C = 256
cl = torch.channels_last
N = int(os.environ.get("FR_AB_N", "4"))
variants = tuple(int(x) for x in os.environ.get("FR_AB", "0,9,6,3").split(","))
for H, stride in ((128, 8), (64, 16)):
    nset = 3 if H == 128 else 10
    sets = [tuple(torch.randn(N, C, H, H, device=dev).contiguous(memory_format=cl) for _ in range(4)) for _ in range(nset)]
    ba, bb = torch.randn(C, device=dev), torch.randn(C, device=dev)
    boxes = syn.fr_level_boxes(N, H, H, stride, 3, device=dev)
    ref = None
    for rep in range(2):
        for v in variants:
            _C.set_option("fr_dbg", v)
            a, b, r, o = sets[0]
            fr_module_nhwc(a, b, ba, bb, r, boxes, 1.0 / stride, 1, o)
            if ref is None: ref = o.clone()
            same = bool(torch.equal(o, ref))
            for i in range(4):
                a, b, r, o = sets[i % nset]; fr_module_nhwc(a, b, ba, bb, r, boxes, 1.0 / stride, 1, o)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for i in range(30):
                a, b, r, o = sets[i % nset]; fr_module_nhwc(a, b, ba, bb, r, boxes, 1.0 / stride, 1, o)
            e.record(); torch.cuda.synchronize()
            print(f"H={H} N={N} fr_dbg={v}: {s.elapsed_time(e) * 1000 / 30:6.1f} us  bit-equal: {same}", flush=True)
    del sets
_C.set_option("fr_dbg", 0)
