"""Reads the per-phase cycle counters of an instrumented build of nms_reduce_pipe_kernel
(tools/probes/abl/libprof.so, made by hand from a patched copy of r3_nms.hip; never committed).
Phases per 64-row block, wave 0 and wave 1: nz ring store + request | stage 1 (scan) | barrier 1 |
stage 2 (OR words) | word address + request | barrier 2."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from r3det import _C  # noqa: E402
_C.LIB_PATH = os.path.join(ROOT, "tools", "probes", "abl", "libprof.so")
from r3det import synthetic as syn  # noqa: E402
from r3det.ops import batched_rnms  # noqa: E402

L = _C.lib()
dev = torch.device("cuda")
for n in (2000, 5344, 8576):
    mb, ms = syn.nms_pool(n * 10 // 6 + 64, 77 + n, device=dev)
    sc, lab = ms[:, :-1].max(1)
    idx = torch.nonzero(sc > 0.05).squeeze(1)[:n]
    b, s, l = mb[idx].contiguous(), sc[idx].contiguous(), lab[idx].contiguous()
    for _ in range(3):
        batched_rnms(b, s, l, 0.1)
    torch.cuda.synchronize()
    out = (ctypes.c_longlong * 16)()
    L.r3det_debug_reduce_prof(out)
    for w, base in (("wave0", 0), ("wave1", 8)):
        v = list(out[base:base + 7])
        blocks = max(1, v[6])
        names = ["nzstore", "stage1", "bar1", "stage2", "wordreq", "bar2"]
        print(f"n={b.size(0)} {w} blocks={blocks} cycles/block (100 MHz ticks x?): " +
              "  ".join(f"{nm} {v[i] / blocks:7.1f}" for i, nm in enumerate(names)) + f"   total {sum(v[:6]) / blocks:8.1f}")
