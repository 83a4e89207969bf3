"""The whole inference step as one HIP graph (GraphedStep) on the bench model: does it record, what does a replay cost
against the eager step and against the dense-only graph + list NMS?   WS_SPREAD=0: the one-label pool."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402
from r3det.models.detectors import GraphedDense, GraphedStep  # noqa: E402

dev = torch.device("cuda", 0)
torch.backends.cudnn.benchmark = True
model, img = bench.build_model(dev, 100, spread=os.environ.get("WS_SPREAD", "1") == "1")
print("model built", flush=True)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


eager = model.simple_test(img)
print("eager kept", [int(d.size(0)) for d, _ in eager], flush=True)
print(f"eager step        {timeit(lambda: bench.model_step(model, img)):7.3f} ms", flush=True)
gd = GraphedDense(model, img)
print(f"dense graph + list NMS {timeit(lambda: gd.simple_test(img)):7.3f} ms", flush=True)
g = GraphedStep(model, img)
print("whole step recorded, capacity", g.nms.cap, flush=True)
out, redo = g.step(img)
torch.cuda.synchronize()
print("first replay: capacity overflow of the step before =", redo, "; kept", [int(c) for c in out[:, -1, 0].tolist()], flush=True)
print(f"whole-step graph  {timeit(lambda: g.step(img)):7.3f} ms", flush=True)
res = g.nms.lists()
# (two passes of the network are not bit-identical on this stack -- MIOpen's convolutions --, so the comparison is on the
# kept counts only (near-tied scores change places between two passes); the graph's NMS against the list form on ONE dense output is
# tests/test_model.py's bit-exact check)
same_counts = [int(a[0].size(0)) == int(b[0].size(0)) for a, b in zip(eager, res)]
print("kept counts equal to the eager step's:", all(same_counts))
