"""The per-level pre-NMS pool (r3det_level_pool: keys + select + emit) at the shapes of the two BASELINE models, for
`rocprofv3 --kernel-trace`: R3Det's refine head (A = 1, N = 4: levels of 16 384 and 4096 rows are cut at 2000) and
RRetinaNet's head (A = 9, N = 2: 147 456 / 36 864 / 9216 rows)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from r3det.ops import fr_boxes  # noqa: E402

dev = torch.device("cuda")
C, k = 15, 2000
g = torch.Generator(device="cuda").manual_seed(1)
for name, N, A, sizes in (("r3det-refine", 4, 1, (128, 64)), ("rretinanet", 2, 9, (128, 64, 32))):
    for H in sizes:
        L = H * H * A
        cls = torch.randn(N, A * C, H, H, device=dev, generator=g) * 1.5 - 4.0   # detector-like: few confident rows
        cls = cls.contiguous(memory_format=torch.channels_last)
        reg = (torch.randn(N, A * 5, H, H, device=dev, generator=g) * 0.2).contiguous(memory_format=torch.channels_last)
        anchors = torch.rand(L, 5, device=dev) * 50 + 5
        boxes = torch.empty(N, k, 5, device=dev)
        scores = torch.empty(N, k, C + 1, device=dev)
        for _ in range(3):
            fr_boxes.level_pool(cls, reg, anchors, A, C, k, (1024, 1024), boxes, scores, 0)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            fr_boxes.level_pool(cls, reg, anchors, A, C, k, (1024, 1024), boxes, scores, 0)
        e.record()
        torch.cuda.synchronize()
        print(f"{name:13s} N={N} A={A} {H:3d}x{H:<3d} rows {L:6d}: {s.elapsed_time(e) * 50:7.1f} us per level_pool call", flush=True)

# all five levels of a head in one call (r3det_levels_pool): what the models run
for name, N, A in (("r3det-refine", 4, 1), ("rretinanet", 2, 9)):
    sizes = (128, 64, 32, 16, 8)
    cls = [(torch.randn(N, A * C, H, H, device=dev, generator=g) * 1.5 - 4.0).contiguous(memory_format=torch.channels_last) for H in sizes]
    reg = [(torch.randn(N, A * 5, H, H, device=dev, generator=g) * 0.2).contiguous(memory_format=torch.channels_last) for H in sizes]
    anchors = [torch.rand(H * H * A, 5, device=dev) * 50 + 5 for H in sizes]
    n = sum(min(k, H * H * A) for H in sizes)
    boxes = torch.empty(N, n, 5, device=dev)
    scores = torch.empty(N, n, C + 1, device=dev)
    for _ in range(3):
        fr_boxes.levels_pool(cls, reg, anchors, A, C, k, (1024, 1024), boxes, scores)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        fr_boxes.levels_pool(cls, reg, anchors, A, C, k, (1024, 1024), boxes, scores)
    e.record()
    torch.cuda.synchronize()
    print(f"{name:13s} N={N} A={A} five levels, {n} pool rows: {s.elapsed_time(e) * 50:7.1f} us per levels_pool call", flush=True)
