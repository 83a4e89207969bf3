"""The NCHW FR backward gather alone (index built once) on rotating buffers: the driver for PMC passes of
frn_gather_kernel.  FR_BWD_N (4), FR_BWD_FIELD=regular|adversarial|trained."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from r3det import synthetic as syn  # noqa: E402
from r3det.ops.feature_refine import fr_backward_index, fr_backward_indexed  # noqa: E402

dev = torch.device("cuda")
N, C, H, stride = int(os.environ.get("FR_BWD_N", 4)), 256, 128, 8
field = os.environ.get("FR_BWD_FIELD", "regular")
boxes = syn.fr_level_boxes(N, H, H, stride, 3, device=dev)
if field == "adversarial":
    boxes[:, :2] = torch.rand(boxes.shape[0], 2, device=dev) * (H * stride)
elif field == "trained":
    g = (boxes[:, :2] / (4 * stride)).floor() * (4 * stride) + 2 * stride
    boxes[:, :2] = g + torch.randn_like(g) * 0.3 * stride
nset = max(3, int(0.9e9 // (2 * N * C * H * H * 4)))
sets = [tuple(torch.randn(N, C, H, H, device=dev) for _ in range(2)) for _ in range(nset)]
ix = fr_backward_index(boxes, N, C, H, H, 1.0 / stride, 1)
for i in range(20):
    g, o = sets[i % nset]
    fr_backward_indexed(g, 1, o, ix)
torch.cuda.synchronize()
