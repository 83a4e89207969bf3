"""Fuzz of the batched multiclass entry (multiclass_nms_rotated_batch: r3det_mcnms_select + r3det_mcnms*): nms_impl 6
(sorted chunks forced) against 7 (counting form) on random batches -- 1 .. 4 images, ragged candidate counts (an image
may have none), 1 .. 20 classes, v1 / v2 / v3, spread and clustered boxes.   python tools/mcnms_fuzz.py [cases] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from r3det import _C  # noqa: E402
from r3det.core.post_processing import multiclass_nms_rotated_batch  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(cases):
    B = int(rng.integers(1, 5))
    n = int(rng.choice([rng.integers(10, 300), rng.integers(300, 3000), rng.integers(3000, 9000)]))
    K = int(rng.integers(1, 21))
    typ = str(rng.choice(["v1", "v2", "v3"]))
    span = float(rng.choice([200.0, 1000.0]))
    if rng.random() < 0.5:
        xy = rng.uniform(0, span, (B, n, 2))
    else:
        c = rng.uniform(0, span, (B, max(1, n // 30), 2))
        idx = rng.integers(0, c.shape[1], (B, n))
        xy = np.take_along_axis(c, idx[..., None].repeat(2, -1), 1) + rng.normal(0, 5.0, (B, n, 2))
    wh = rng.uniform(4, 90, (B, n, 2))
    th = rng.uniform(-1.5, 1.5, (B, n, 1))
    boxes = torch.from_numpy(np.concatenate([xy, wh, th], -1).astype(np.float32)).cuda()
    sc = rng.uniform(0, 1, (B, n, K + 1)).astype(np.float32) ** 3
    if rng.random() < 0.5:
        sc = np.round(sc, 2)
    for b in range(B):
        if rng.random() < 0.2:
            sc[b] = 0.0  # an image without candidates
    scores = torch.from_numpy(sc).cuda()
    cfg = dict(type=typ, iou_thr=float(rng.choice([0.1, 0.3])))
    max_num = int(rng.choice([2000, 100]))
    got = {}
    for impl in (7, 6):
        _C.set_option("nms_impl", impl)
        got[impl] = multiclass_nms_rotated_batch(boxes, scores, 0.05, cfg, max_num)
    _C.set_option("nms_impl", 0)
    same = all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(got[6], got[7]))
    if not same:
        bad += 1
        print(f"MISMATCH case {case}: B={B} n={n} K={K} {typ} max_num={max_num}", flush=True)
print("mismatches:", bad, "of", cases)
sys.exit(1 if bad else 0)
