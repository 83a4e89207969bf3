"""Does the producer of the FR input change the FR kernel's duration?  level 0, N = 4."""
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
x_cl = x.contiguous(memory_format=torch.channels_last)
big = torch.empty(512 * 1024 * 1024 // 4, device=dev)  # 512 MB: flushes L2 / Infinity Cache when written


def measure(producer, name, reps=30):
    ts = []
    for _ in range(reps):
        y = producer()
        o = torch.empty_like(y)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fr_forward(y, b, 1 / 8, 1, o)
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    ts.sort()
    print(f"{name:44s} FR med {ts[len(ts) // 2]:6.1f} us  min {ts[0]:6.1f} us", flush=True)


measure(lambda: x, "same NCHW tensor every time (cache-warm)")
measure(lambda: x + 0, "fresh NCHW tensor from an elementwise add")
measure(lambda: x_cl.contiguous(), "fresh NCHW tensor from channels_last copy")
measure(lambda: (big.zero_(), x)[1], "same tensor after a 512 MB cache flush")
measure(lambda: (x + 0, big.zero_())[0], "fresh tensor, then cache flush")
