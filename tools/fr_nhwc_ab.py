"""A/B of the two channels_last sampler kernels (option fr_dbg: 0 deep pipeline on 8 x 8 tiles, 2 one-step
pipeline on 4 x 4 tiles) at several batch sizes (the workgroup count per CU changes the tail)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# launch variants that were not shipped / clock stamps live in the probes build of the library (make probes)
os.environ.setdefault("R3DET_HIP_LIB", os.path.join(ROOT, "r3det-pytorch_amd", "libr3det_hip_probes.so"))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")): sys.path.insert(0, p)
import torch
from r3det import synthetic as syn, _C
from r3det.ops.feature_refine import fr_module_nhwc
dev = torch.device("cuda")
C, H = 256, 128
cl = torch.channels_last
for N in tuple(int(x) for x in os.environ.get("FR_AB_N", "4,8,16").split(",")):
    nset = 3 if N <= 8 else 2
    sets = [tuple(torch.randn(N, C, H, H, device=dev).contiguous(memory_format=cl) for _ in range(4)) for _ in range(nset)]
    ba, bb = torch.randn(C, device=dev), torch.randn(C, device=dev)
    boxes = syn.fr_level_boxes(N, H, H, 8, 3, device=dev)
    for dbg in tuple(int(x) for x in os.environ.get("FR_AB", "2,0").split(",")):
        _C.set_option("fr_dbg", dbg)
        for i in range(4):
            a, b, r, o = sets[i % nset]; fr_module_nhwc(a, b, ba, bb, r, boxes, 0.125, 1, o)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(20):
            a, b, r, o = sets[i % nset]; fr_module_nhwc(a, b, ba, bb, r, boxes, 0.125, 1, o)
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1000 / 20
        alg = 16 * N * C * H * H + 20 * N * H * H
        print(f"N={N:2d} fr_dbg={dbg}: {us:7.1f} us  {alg / us / 1e3:7.1f} GB/s", flush=True)
    del sets
_C.set_option("fr_dbg", 0)
