"""A/B of the NCHW cell sampler's non-temporal plane loads / stores (option fr_dbg 21: plain loads and stores),
levels 0 / 1, N = 4, buffers rotating beyond the Infinity Cache; both forms bit-identical."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# launch variants that were not shipped / clock stamps live in the probes build of the library (make probes)
os.environ.setdefault("R3DET_HIP_LIB", os.path.join(ROOT, "r3det-pytorch_amd", "libr3det_hip_probes.so"))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")): sys.path.insert(0, p)
import torch
from r3det import _C, synthetic as syn
from r3det.ops.feature_refine import fr_forward
dev = torch.device("cuda")
N, C = 4, 256
for lvl, H in ((0, 128), (1, 64)):
    st = syn.STRIDES[lvl]
    nset = 5 if H == 128 else 20
    sets = [(torch.randn(N, C, H, H, device=dev), torch.empty(N, C, H, H, device=dev)) for _ in range(nset)]
    boxes = syn.fr_level_boxes(N, H, H, st, 3, device=dev)
    ref = None
    for rep in range(3):
        for dbg in (0, 21):
            _C.set_option("fr_dbg", dbg)
            f, o = sets[0]
            fr_forward(f, boxes, 1.0 / st, 1, o)
            if ref is None: ref = o.clone()
            same = bool(torch.equal(o, ref))
            for i in range(5):
                f, o = sets[i % nset]; fr_forward(f, boxes, 1.0 / st, 1, o)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for i in range(30):
                f, o = sets[i % nset]; fr_forward(f, boxes, 1.0 / st, 1, o)
            e.record(); torch.cuda.synchronize()
            print(f"level {lvl} N={N} {'non-temporal' if dbg == 0 else 'plain       '}: {s.elapsed_time(e) * 1000 / 30:6.1f} us  bit-equal: {same}", flush=True)
    del sets
_C.set_option("fr_dbg", 0)
