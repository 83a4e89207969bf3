"""Does the distance between the four streams of the module-tail launch (a, b, res, out; 64 MiB each at level 0,
N = 4) matter?  The tensors are carved out of ONE allocation at controlled byte distances (64 MiB + delta), three
rotating sets; jittered bench field and a field without taps."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")): sys.path.insert(0, p)
import torch
from r3det import synthetic as syn
from r3det.ops.feature_refine import fr_module_nhwc
dev = torch.device("cuda")
N, C, H = 4, 256, 128
n = N * C * H * H
ba, bb = torch.randn(C, device=dev), torch.randn(C, device=dev)
boxes = syn.fr_level_boxes(N, H, H, 8, 3, device=dev)
far = boxes.clone(); far[:, 0] = -1000; far[:, 1] = -1000
for delta in (0, 256, 4096, 65536 + 4096, 1 << 20, (1 << 20) + 4096 * 13, (3 << 20) + 256 * 7):
    step = n + delta // 4
    big = torch.randn(12 * step + 16, device=dev)
    sets = [tuple(big[(4 * s + k) * step:(4 * s + k) * step + n].view(N, H, H, C).permute(0, 3, 1, 2) for k in range(4)) for s in range(3)]
    assert sets[0][0].is_contiguous(memory_format=torch.channels_last)
    res = []
    for bx in (boxes, far):
        for i in range(6):
            a, b, r, o = sets[i % 3]; fr_module_nhwc(a, b, ba, bb, r, bx, 0.125, 1, o)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(30):
            a, b, r, o = sets[i % 3]; fr_module_nhwc(a, b, ba, bb, r, bx, 0.125, 1, o)
        e.record(); torch.cuda.synchronize()
        res.append(s.elapsed_time(e) * 1000 / 30)
    print(f"stream distance 64 MiB + {delta:8d} B: jittered field {res[0]:6.1f} us   no taps {res[1]:6.1f} us", flush=True)
    del sets, big
