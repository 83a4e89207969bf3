"""FeatureRefine forward + backward (autograd) on the pyramid of the training step (N = 4, C = 256):
microseconds per level, with the backward's packing made at forward time (side stream) or inside
the backward call."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402

from r3det import synthetic as syn  # noqa: E402
import r3det.ops.feature_refine  # noqa: E402,F401

FRM = sys.modules['r3det.ops.feature_refine']

feats, boxes = syn.fr_pyramid(4, 256, 9, device="cuda")


A = torch.randn(4096, 4096, device="cuda")


def step(x, g, lvl, s, filler):
    x.grad = None
    y = FRM.feature_refine(x, boxes[lvl], s, 1)
    for _ in range(filler):  # the rest of the step between the sampler's forward and its backward
        torch.mm(A, A)
    y.backward(g)


def run(lvl, reps, filler):
    x = feats[lvl].clone().requires_grad_(True)
    g = torch.randn_like(x)
    s = 1.0 / syn.STRIDES[lvl]
    for _ in range(3):
        step(x, g, lvl, s, filler)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        step(x, g, lvl, s, filler)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


FRM.PACK_AT_FORWARD = True
orig = FRM.fr_backward_prepare_async
for filler in (0, 4):
    for lvl in (0, 1):
        res = []
        for _ in range(2):
            FRM.fr_backward_prepare_async = orig
            t1 = run(lvl, 30, filler)
            FRM.fr_backward_prepare_async = lambda *a, **k: None
            t2 = run(lvl, 30, filler)
            res.append((t1, t2))
        t1, t2 = min(r[0] for r in res), min(r[1] for r in res)
        print(f"level {lvl}, {filler} filler GEMMs: step {t1:8.1f} us with the packing at forward time, "
              f"{t2:8.1f} us inside the backward", flush=True)
