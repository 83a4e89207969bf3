"""The whole R3Det inference step (bench.py's model, batch 4 x 1024^2) with the channels_last sampler's two forms:
option fr_dbg 9 (4 x 4 tile pairs) / 0 (wide regions, the default), alternating, same process."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# launch variants that were not shipped / clock stamps live in the probes build of the library (make probes)
os.environ.setdefault("R3DET_HIP_LIB", os.path.join(ROOT, "r3det-pytorch_amd", "libr3det_hip_probes.so"))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")): sys.path.insert(0, p)
import torch
import bench
from r3det import _C
dev = torch.device("cuda", 0); torch.cuda.set_device(dev); torch.backends.cudnn.benchmark = True
model, img = bench.build_model(dev, 100)
def step():
    with torch.no_grad():
        return model.simple_test(img)
for _ in range(5): step()
torch.cuda.synchronize()
for rep in range(3):
    for dbg in (9, 0):
        _C.set_option("fr_dbg", dbg)
        for _ in range(3): step()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(20): step()
        torch.cuda.synchronize()
        print(f"fr_dbg {dbg}: {(time.perf_counter() - t) / 20 * 1e3:7.3f} ms per step", flush=True)
_C.set_option("fr_dbg", 0)
