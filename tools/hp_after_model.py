"""Per-iteration wall time of the hot-path step right after a model run in the same process
(first process on a fresh box vs later ones)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda")
torch.backends.cudnn.benchmark = True
if "--model" in sys.argv:
    model, img = bench.build_model(dev, 100)
    for _ in range(3):
        bench.model_step(model, img)
    torch.cuda.synchronize()
    del model, img
    torch.cuda.empty_cache()
wl = bench.build_hot_workload(dev, 7)
ts = []
for i in range(30):
    torch.cuda.synchronize()
    t = time.perf_counter()
    bench.hot_path_step(wl)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t) * 1e3)
print("hot path ms per iteration:", " ".join(f"{t:.2f}" for t in ts))
