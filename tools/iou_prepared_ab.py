"""Plain vs prepared-columns overlap matrix and assignment (same process, alternating): 128 x 196 416 (the anchor grid),
128 x 21 824, v1 and v3."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from r3det import _C, synthetic as syn  # noqa: E402
from r3det.core.bbox.assigners import MaxIoUAssigner  # noqa: E402
from r3det.ops.iou import GEOM, prepare_columns  # noqa: E402

dev = torch.device("cuda")
L = _C.lib()
gt = syn.dota_like_rboxes(128, 5, device=dev)
grid = syn.anchor_grid(device=dev)


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1000 / reps


for version in ("v1", "v3"):
    for cols in (grid, grid[-21824:].contiguous()):
        n1, n2 = gt.size(0), cols.size(0)
        geom = GEOM[version]
        prep = prepare_columns(cols, version)
        ws, wsb = _C.iou_workspace(n1, n2, dev)
        out = torch.empty(n1, n2, device=dev)
        plain_fn = L.r3det_rbbox_geo_mat_iou_iof if geom == 1 else L.r3det_box_iou_rotated_overlaps
        mode = 0 if geom == 1 else 1
        plain = lambda: plain_fn(_C.ptr(gt), n1, _C.ptr(cols), n2, mode, _C.ptr(out), _C.ptr(ws), wsb, _C.stream())  # noqa: E731
        prepared = lambda: L.r3det_iou_mat_prepared(geom, _C.ptr(gt), n1, _C.ptr(cols), n2, _C.ptr(prep), mode, _C.ptr(out),  # noqa: E731
                                                    _C.ptr(ws), wsb, _C.stream())
        for rnd in range(2):
            print(f"{version} {n1} x {n2}: plain {timed(plain):6.1f} us   prepared {timed(prepared):6.1f} us", flush=True)
        print(f"   (prepare_columns itself: {timed(lambda: prepare_columns(cols, version)):6.1f} us)")
a = MaxIoUAssigner(0.5, 0.4, 0., iou_calculator=dict(type='RBboxOverlaps2D_v1'))
for rnd in range(2):
    print(f"assign 128 x 196416: plain {timed(lambda: a.assign(grid, gt)):6.1f} us   prepared "
          f"{timed(lambda: a.assign(grid, gt, shared_key='g')):6.1f} us")
