"""Backbone + FPN + heads forward time: NCHW vs channels_last (fp32, MIOpen find mode)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from r3det.models import R3Det  # noqa: E402

torch.backends.cudnn.benchmark = True
dev = torch.device("cuda")
torch.manual_seed(0)
model = R3Det().eval().to(dev)
img = torch.randn(4, 3, 1024, 1024, device=dev)


def run(m, x, what):
    with torch.no_grad():
        for _ in range(3):
            feats = m.neck(m.backbone(x))
            outs = m.bbox_head(feats)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(10):
            feats = m.neck(m.backbone(x))
        torch.cuda.synchronize()
        t1 = (time.perf_counter() - t) / 10
        t = time.perf_counter()
        for _ in range(10):
            outs = m.bbox_head(feats)
        torch.cuda.synchronize()
        t2 = (time.perf_counter() - t) / 10
    print(f"{what}: backbone+neck {t1 * 1e3:.2f} ms   first head {t2 * 1e3:.2f} ms", flush=True)


run(model, img, "NCHW")
model_cl = model.to(memory_format=torch.channels_last)
run(model_cl, img.contiguous(memory_format=torch.channels_last), "channels_last")
