"""Times fr_forward (FR_IMPL env, default 10 = cell) with alternative builds of the library.

    python tools/fr_chan_ablate.py tools/probes/abl/lib0.so tools/probes/abl/lib1.so ...

Each library runs in its own child process (a ctypes library cannot be swapped in place).
The ablation builds are made by hand from a patched copy of r3_fr.hip (see DESIGN_HISTORY.md 4.3);
they give wrong results on purpose and are never committed.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {root!r} + "/r3det-pytorch_amd")
import torch
from r3det import _C
_C.LIB_PATH = {lib!r}
from r3det import synthetic as syn
from r3det.ops.feature_refine import fr_forward
_C.lib(); _C.set_option("fr_impl", int(__import__("os").environ.get("FR_IMPL", "10")))
for N in (4, 16):
    feats, boxes = syn.fr_pyramid(N, 256, 9, device=torch.device("cuda"))
    f, b = feats[0], boxes[0]
    o = torch.empty_like(f)
    for _ in range(5): fr_forward(f, b, 1 / 8, 1, o)
    ts = []
    for _ in range(7):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): fr_forward(f, b, 1 / 8, 1, o)
        e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3 / 20)
    ts.sort()
    print("  N=%d  med %.1f us  min %.1f us  (%.0f GB/s alg)" % (N, ts[3], ts[0], (8 * f.numel() + 20 * b.size(0)) / ts[3] / 1e3))
"""

if __name__ == "__main__":
    for lib in sys.argv[1:]:
        print(lib, flush=True)
        subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT, lib=os.path.abspath(lib))], check=False)
