"""A few R3Det training steps (BASELINE configs[4]: batch 2 x 1024^2, 128 GT per image) for rocprofv3
--kernel-trace; prints the step time of the profiled run."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
tr = bench.build_train(dev, 300, 1)
for _ in range(3):
    bench.train_step(tr)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(3):
    bench.train_step(tr)
torch.cuda.synchronize()
print(f"train step: {(time.perf_counter() - t) / 3 * 1e3:.3f} ms", flush=True)
