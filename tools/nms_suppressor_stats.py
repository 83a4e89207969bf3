"""How many suppressors (boxes of the same class with a higher score and IoU > thr) the rows of tools/nms_prof.py's
pools have: the reducer keeps the first 32 of a row in its list and the rest as bits of a dense mask row."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from r3det import synthetic as syn  # noqa: E402
from r3det.ops import rbbox_iou as rbbox_overlaps  # noqa: E402

dev = torch.device("cuda")
for n in [int(x) for x in (sys.argv[1:] or ["8576", "32768"])]:
    mb, ms = syn.nms_pool(n * 10 // 6 + 64, 77 + n, device=dev)
    sc, lab = ms[:, :-1].max(1)
    idx = torch.nonzero(sc > 0.05).squeeze(1)[:n]
    b, s, l = mb[idx].contiguous(), sc[idx].contiguous(), lab[idx].contiguous()
    order = torch.argsort(s, descending=True, stable=True)
    b, l = b[order], l[order]
    cnt = torch.zeros(n, dtype=torch.int64, device=dev)
    for i0 in range(0, n, 4096):
        iou = rbbox_overlaps(b[i0:i0 + 4096], b)                      # rows i0.. against all
        same = l[i0:i0 + 4096, None] == l[None, :]
        earlier = torch.arange(n, device=dev)[None, :] < torch.arange(i0, min(i0 + 4096, n), device=dev)[:, None]
        cnt[i0:i0 + 4096] = ((iou > 0.1) & same & earlier).sum(1)
    c = cnt.cpu()
    print(f"n={n}: rows with >0 suppressors {(c > 0).sum().item()}, >8 {(c > 8).sum().item()}, >32 {(c > 32).sum().item()}, "
          f">64 {(c > 64).sum().item()}, >128 {(c > 128).sum().item()}, max {c.max().item()}, mean {c.float().mean().item():.1f}; "
          f"rows per class: max {torch.bincount(l.cpu()).max().item()}")
