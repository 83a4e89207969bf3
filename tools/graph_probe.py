"""Does the network + decoding part of the step capture into a HIP graph, and what does it buy?"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402
from r3det.models.detectors import GraphedDense  # noqa: E402

dev = torch.device("cuda", 0)
torch.backends.cudnn.benchmark = True
model, img = bench.build_model(dev, 100)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


eager = [r[0].clone() for r in model.simple_test(img)]
print(f"eager  step {timeit(lambda: model.simple_test(img)):7.3f} ms   dense only {timeit(lambda: model.dense_test(img)):7.3f} ms", flush=True)
g = GraphedDense(model, img)
print(f"graph  step {timeit(lambda: g.simple_test(img)):7.3f} ms   dense only {timeit(lambda: g(img)):7.3f} ms", flush=True)
out = g.simple_test(img)
print("same detections:", all(torch.equal(a, b[0]) for a, b in zip(eager, out)), [int(b[0].size(0)) for b in out])
