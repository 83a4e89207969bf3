import os, sys
ROOT="/root/repo"
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")): sys.path.insert(0, p)
import torch, time
import bench
from r3det import _C
dev = torch.device("cuda", 0); torch.cuda.set_device(dev); torch.backends.cudnn.benchmark = True
model, img = bench.build_model(dev, 100)
x = model.extract_feat(img)
with torch.no_grad():
    boxes, scores = model.dense_test(img)
print("pool", boxes.shape, "cand/img", [(scores[i,:,:-1] > 0.05).sum().item() for i in range(4)], "box max", boxes.max().item(), "w max", boxes[...,2].max().item(), "h max", boxes[...,3].max().item())
from r3det.core.post_processing import multiclass_nms_rotated_batch, CapacityHint
h = CapacityHint()
for impl in (0, 2, 0, 2):
    _C.set_option("nms_impl", impl)
    for _ in range(3): multiclass_nms_rotated_batch(boxes, scores, 0.05, dict(iou_thr=0.1), 2000, hint=h)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(20): multiclass_nms_rotated_batch(boxes, scores, 0.05, dict(iou_thr=0.1), 2000, hint=h)
    torch.cuda.synchronize(); print("nms_impl", impl, (time.perf_counter()-t)/20*1e6, "us")
_C.set_option("nms_impl", 0)
