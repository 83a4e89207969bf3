"""Does the distance between the FR input and output buffers matter?  level 0, N = 4 (67 MB each)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from r3det import synthetic as syn  # noqa: E402
from r3det.ops.feature_refine import fr_forward  # noqa: E402

dev = torch.device("cuda")
feats, boxes = syn.fr_pyramid(4, 256, 9, device=dev)
x, b = feats[0], boxes[0]
nel = x.numel()
pool = torch.empty(nel * 4 + (64 << 20), device=dev)  # floats
inp = pool[:nel].view_as(x)
inp.copy_(x)
for off_bytes in (0, 4096, 65536, 1 << 20, 2 << 20, 3 << 20, 16 << 20, 33 << 20, (64 << 20) + 4096, 128 << 20,
                  (128 << 20) + (1 << 20) + 12288):
    start = nel + off_bytes // 4
    out = pool[start:start + nel].view_as(x)
    for _ in range(5):
        fr_forward(inp, b, 1 / 8, 1, out)
    ts = []
    for _ in range(7):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            fr_forward(inp, b, 1 / 8, 1, out)
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3 / 20)
    ts.sort()
    print(f"out = in + 64 MiB + {off_bytes:>10d} B: med {ts[3]:6.1f} us  min {ts[0]:6.1f} us", flush=True)
