"""The channels_last FeatureRefineModule tail (r3det_feature_refine_module_nhwc) at level 0 of a 1024^2 input
(N = 4, C = 256, 128 x 128), launched over three rotating buffer sets (3 x 268 MB: beyond the 256 MiB Infinity
Cache) -- the driver for `rocprofv3 --kernel-trace` and the PMC passes of the roofline kernel
(tools/pmc_groups.sh ... tools/fr_nhwc_prof.py)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from r3det import synthetic as syn  # noqa: E402
from r3det.ops.feature_refine import fr_module_nhwc  # noqa: E402

from r3det import _C  # noqa: E402

_C.set_option("fr_dbg", int(os.environ.get("FR_DBG", "0")))  # A/B variants of the launch (csrc/r3_fr.hip)
if "FR_WALK" in os.environ:
    _C.set_option("fr_walk", int(os.environ["FR_WALK"]))  # strip height of the tile-pair walk (0: row-major)
dev = torch.device("cuda")
N, C, H = int(os.environ.get("FR_N", "4")), 256, 128
cl = torch.channels_last
boxes = syn.fr_level_boxes(N, H, H, 8, 3, device=dev)
NSET = max(3, 12 // N)
sets = [tuple(torch.randn(N, C, H, H, device=dev).contiguous(memory_format=cl) for _ in range(4)) for _ in range(NSET)]
ba, bb = torch.randn(C, device=dev), torch.randn(C, device=dev)
for i in range(6):
    a, b, r, o = sets[i % NSET]
    fr_module_nhwc(a, b, ba, bb, r, boxes, 0.125, 1, o)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for i in range(30):
    a, b, r, o = sets[i % NSET]
    fr_module_nhwc(a, b, ba, bb, r, boxes, 0.125, 1, o)
e.record()
torch.cuda.synchronize()
us = s.elapsed_time(e) * 1000 / 30
alg = 16 * N * C * H * H + 20 * N * H * H
print(f"fr_module_nhwc level 0: {us:7.1f} us per launch, {alg / us / 1e3:7.1f} GB/s on {alg} algorithmic bytes", flush=True)
