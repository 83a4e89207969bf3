"""A/B of the tile-pair walk (option fr_walk: strip height of r3_fr_tap.h's pair_walk, 0 = row-major over the
triangle) for the channels_last sampler forward (fr_module_nhwc) and backward gather, level 0 / 1, rotating buffers."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")): sys.path.insert(0, p)
import torch
from r3det import synthetic as syn, _C
from r3det.ops.feature_refine import fr_module_nhwc, fr_backward_nhwc
dev = torch.device("cuda")
C = 256
cl = torch.channels_last
N = int(os.environ.get("FR_AB_N", "4"))
walks = tuple(int(x) for x in os.environ.get("FR_WALK", "0,2,4,8,16,32").split(","))
for H, stride in ((128, 8), (64, 16)):
    nset = 3 if H == 128 else 10
    sets = [tuple(torch.randn(N, C, H, H, device=dev).contiguous(memory_format=cl) for _ in range(4)) for _ in range(nset)]
    ba, bb = torch.randn(C, device=dev), torch.randn(C, device=dev)
    boxes = syn.fr_level_boxes(N, H, H, stride, 3, device=dev)
    ref_f = ref_b = None
    for wk in walks:
        _C.set_option("fr_walk", wk)
        a, b, r, o = sets[0]
        fr_module_nhwc(a, b, ba, bb, r, boxes, 1.0 / stride, 1, o)
        g = torch.empty_like(a); fr_backward_nhwc(a, boxes, 1.0 / stride, 1, g, overwrite=True)
        if ref_f is None: ref_f, ref_b = o.clone(), g.clone()
        same = bool(torch.equal(o, ref_f)) and bool(torch.equal(g, ref_b))
        res = []
        for fn in ("fwd", "bwd"):
            for i in range(4):
                a, b, r, o = sets[i % nset]
                fr_module_nhwc(a, b, ba, bb, r, boxes, 1.0 / stride, 1, o) if fn == "fwd" else fr_backward_nhwc(a, boxes, 1.0 / stride, 1, o, overwrite=True)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for i in range(30):
                a, b, r, o = sets[i % nset]
                fr_module_nhwc(a, b, ba, bb, r, boxes, 1.0 / stride, 1, o) if fn == "fwd" else fr_backward_nhwc(a, boxes, 1.0 / stride, 1, o, overwrite=True)
            e.record(); torch.cuda.synchronize()
            res.append(s.elapsed_time(e) * 1000 / 30)
        print(f"H={H} N={N} fr_walk={wk:2d}: forward {res[0]:6.1f} us  backward(index+gather) {res[1]:6.1f} us  bit-equal to walk 0: {same}", flush=True)
    del sets
