"""How many pairs pass the stream kernel's conservative test, how many of them overlap, and how they are
spread over the (16-row x 1024-column) tiles -- for the three IoU shapes of the bench.
    python tools/iou_survivors.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from r3det import _C, synthetic as syn  # noqa: E402

dev = torch.device("cuda")
L = _C.lib()
anchors = syn.anchor_grid(device=dev)
gt = syn.dota_like_rboxes(128, 5, device=dev)
refined = torch.cat([syn.fr_level_boxes(1, 1024 // s, 1024 // s, s, 50 + i, device=dev) for i, s in enumerate(syn.STRIDES)])
a, g = syn.rand_rboxes(1000, 0, device=dev), syn.rand_rboxes(128, 1, device=dev)
for name, b1, b2 in (("128x196416", gt, anchors), ("128x21824", gt, refined), ("1000x128", a, g)):
    n1, n2 = b1.size(0), b2.size(0)
    nbytes = int(L.r3det_iou_workspace_bytes(n1, n2))
    ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    out = torch.empty(n1, n2, device=dev)
    _C.check(L.r3det_rbbox_geo_mat_iou_iof(_C.ptr(b1), n1, _C.ptr(b2), n2, 0, _C.ptr(out), _C.ptr(ws), nbytes, _C.stream()), "iou")
    torch.cuda.synchronize()
    surv = int(ws[:4].view(torch.int32)[0])
    nnz = int((out > 0).sum())
    pos = (out > 0)
    # tiles of 16 rows x 1024 cols
    R, Cc = (n1 + 15) // 16, (n2 + 1023) // 1024
    pad = torch.zeros(R * 16, Cc * 1024, dtype=torch.bool, device=dev)
    pad[:n1, :n2] = pos
    per_tile = pad.view(R, 16, Cc, 1024).sum((1, 3)).flatten().float()
    srt = per_tile.sort(descending=True)[0]
    print(f"{name}: pairs {n1 * n2}  survivors {surv} ({100 * surv / (n1 * n2):.2f} %)  overlapping {nnz} "
          f"({100 * nnz / max(1, surv):.1f} % of survivors)  tiles {R * Cc}  nnz/tile max {int(srt[0])} "
          f"p99 {int(srt[int(len(srt) * 0.01)])} median {int(srt[len(srt) // 2])}  "
          f"top-5% tiles hold {100 * float(srt[:max(1, len(srt) // 20)].sum()) / max(1, nnz):.0f} % of nnz", flush=True)
