"""A few RRetinaNet (BASELINE configs[1]) inference steps for rocprofv3 --kernel-trace."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
    sys.path.insert(0, p)
import torch
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
torch.backends.cudnn.benchmark = True
model, img = bench.build_model(dev, 200, "RRetinaNet", bench.RRETINA_BATCH)
for _ in range(6):
    bench.model_step(model, img)
torch.cuda.synchronize()
