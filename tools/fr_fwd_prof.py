"""The plain FR samplers (the reference's operator: out = feat + sample(feat)) per pyramid level at N = 4, C = 256,
both layouts, on rotating buffers where the level is large enough -- the driver for `rocprofv3 --kernel-trace` of
r3det_feature_refine_forward (NCHW: cell / plane kernels + tap table) and r3det_feature_refine_forward_nhwc."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from r3det import synthetic as syn  # noqa: E402
from r3det.ops.feature_refine import fr_forward, fr_forward_nhwc  # noqa: E402

dev = torch.device("cuda")
N, C = int(os.environ.get("FR_FWD_N", 4)), 256
cl = torch.channels_last
feats, boxes = syn.fr_pyramid(N, C, 31, device=dev)
for lay, fn in (("nchw", fr_forward), ("nhwc", fr_forward_nhwc)):
    for lvl, (f, b) in enumerate(zip(feats, boxes)):
        per_set = 2 * 4 * f.numel()
        nset = max(2, min(16, int(6e8 // per_set) + 1))
        sets = []
        for _ in range(nset):
            x = torch.randn_like(f)
            if lay == "nhwc":
                x = x.contiguous(memory_format=cl)
            sets.append((x, torch.empty_like(x)))
        for i in range(2 * nset):
            fn(sets[i % nset][0], b, 1.0 / syn.STRIDES[lvl], 1, sets[i % nset][1])
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(30):
            fn(sets[i % nset][0], b, 1.0 / syn.STRIDES[lvl], 1, sets[i % nset][1])
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1000 / 30
        alg = 8 * f.numel() + 20 * N * f.shape[-1] * f.shape[-2]
        print(f"fr_forward {lay} level {lvl} N={N}: {us:7.1f} us per call, {alg / us / 1e3:7.1f} GB/s on {alg} algorithmic bytes", flush=True)
        del sets
