"""The roofline launch (fr_module_nhwc, level 0, N = 4) ALONE on rotating buffers, but with the box field the bench
model's first stage really produces (its filter_bboxes output at level 0) instead of the synthetic jittered field."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")): sys.path.insert(0, p)
import torch
import bench
from r3det import synthetic as syn
from r3det.ops.feature_refine import fr_module_nhwc
dev = torch.device("cuda", 0); torch.cuda.set_device(dev); torch.backends.cudnn.benchmark = True
model, img = bench.build_model(dev, 100)
with torch.no_grad():
    x = model.extract_feat(img)
    outs = model.bbox_head(x)
    rois = model.bbox_head.filter_bboxes(*outs)
b0 = torch.stack([rois[i][0] for i in range(len(rois))]).reshape(-1, 5).contiguous()
H = 128
print("model field: centres (x, y) of positions (0,0), (0,1), (1,0):", b0[0, :2].tolist(), b0[1, :2].tolist(), b0[H, :2].tolist(),
      " w/h max", b0[:, 2].max().item(), b0[:, 3].max().item())
ys, xs = torch.meshgrid(torch.arange(H, device=dev), torch.arange(H, device=dev), indexing="ij")
d = (b0.view(4, H, H, 5)[..., 0] / 8 - xs) ; e = (b0.view(4, H, H, 5)[..., 1] / 8 - ys)
print("sample offset from the own (transposed) cell, in cells: x mean %.3f std %.3f |max| %.2f ; y mean %.3f std %.3f |max| %.2f" % (d.mean(), d.std(), d.abs().max(), e.mean(), e.std(), e.abs().max()))
del model
torch.cuda.empty_cache()
N, C = 4, 256
cl = torch.channels_last
sets = [tuple(torch.randn(N, C, H, H, device=dev).contiguous(memory_format=cl) for _ in range(4)) for _ in range(3)]
ba, bb = torch.randn(C, device=dev), torch.randn(C, device=dev)
for name, bx in (("bench field", syn.fr_level_boxes(N, H, H, 8, 3, device=dev)), ("model field", b0)):
    for i in range(6):
        a, b, r, o = sets[i % 3]; fr_module_nhwc(a, b, ba, bb, r, bx, 0.125, 1, o)
    torch.cuda.synchronize()
    s, e2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(30):
        a, b, r, o = sets[i % 3]; fr_module_nhwc(a, b, ba, bb, r, bx, 0.125, 1, o)
    e2.record(); torch.cuda.synchronize()
    print(f"{name}: {s.elapsed_time(e2) * 1000 / 30:6.1f} us", flush=True)
