"""Run only the FR forward launches of a 1024^2 R3Det batch (N=4, C=256, 5 levels), for
rocprofv3 --pmc passes:  python3 tools/fr_only.py [reps] [fr_impl]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from r3det import _C, synthetic as syn  # noqa: E402
from r3det.ops.feature_refine import fr_forward  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
impl = int(sys.argv[2]) if len(sys.argv) > 2 else 0
_C.set_option("fr_impl", impl)
dev = torch.device("cuda")
feats, boxes = syn.fr_pyramid(4, 256, 9, device=dev)
outs = [torch.empty_like(f) for f in feats]
for _ in range(reps):
    for f, b, o, s in zip(feats, boxes, outs, syn.STRIDES):
        fr_forward(f, b, 1.0 / s, 1, o)
torch.cuda.synchronize()
print("done")
