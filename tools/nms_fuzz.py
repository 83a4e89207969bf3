"""Fuzz: the batched NMS pipeline's sorted-chunk form (nms_impl 6) against the counting form (7) on random pools --
sizes 1 .. 20 000, 1 .. 40 classes, spread / clustered / degenerate boxes (zero sizes, huge boxes, NaN and inf
coordinates), tied scores -- through the one-pool entries (v1, v3).  Keep lists and rows must be identical.   python tools/nms_fuzz.py [cases] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import r3det.ops.nms as M  # noqa: E402
from r3det import _C  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(cases):
    n = int(rng.choice([rng.integers(1, 200), rng.integers(200, 3000), rng.integers(3000, 20000)]))
    K = int(rng.integers(1, 41))
    kind = rng.choice(["spread", "clusters", "one_pile", "wide"])
    span = float(rng.choice([100.0, 1000.0, 4000.0]))
    if kind == "spread":
        xy = rng.uniform(0, span, (n, 2))
    elif kind == "clusters":
        c = rng.uniform(0, span, (max(1, n // 20), 2))
        xy = c[rng.integers(0, len(c), n)] + rng.normal(0, 6.0, (n, 2))
    elif kind == "one_pile":
        xy = rng.normal(span / 2, 10.0, (n, 2))
    else:
        xy = rng.uniform(0, span, (n, 2))
    wh = rng.uniform(4, 80, (n, 2)) if kind != "wide" else rng.uniform(4, 3 * span, (n, 2))
    th = rng.uniform(-1.6, 1.6, (n, 1))
    b = np.concatenate([xy, wh, th], 1).astype(np.float32)
    if rng.random() < 0.3:
        idx = rng.integers(0, n, max(1, n // 50))
        b[idx, rng.integers(0, 5, len(idx))] = rng.choice([np.nan, np.inf, -np.inf, 0.0, 1e-4, 1e20], len(idx)).astype(np.float32)
    s = rng.uniform(0.05, 1, n).astype(np.float32)
    if rng.random() < 0.5:
        s = np.round(s, 2)
    lab = rng.integers(0, K, n)
    tb, ts, tl = (torch.from_numpy(x).cuda() for x in (b, s, lab))
    for entry in ("r3det_batched_rnms", "r3det_obb_batched_nms"):
        got = {}
        thr = float(rng.choice([0.1, 0.3, 0.5]))
        for impl in (7, 6):
            _C.set_option("nms_impl", impl)
            d, k = M._batched_rnms_device(tb, ts, tl, thr, False, entry=entry)
            got[impl] = (d.clone(), k.clone())
        _C.set_option("nms_impl", 0)
        same = torch.equal(got[6][1], got[7][1]) and torch.equal(got[6][0].nan_to_num(123.0), got[7][0].nan_to_num(123.0))
        if same and kind in ("one_pile", "clusters") and n <= 4000:  # (dense pools: the same answer call after call)
            for _ in range(6):
                d, k = M._batched_rnms_device(tb, ts, tl, thr, False, entry=entry)
                same = same and torch.equal(k, got[7][1])
        if not same:
            bad += 1
            print(f"MISMATCH case {case} {entry} n={n} K={K} kind={kind} span={span} thr={thr}: kept {got[7][1].numel()} vs {got[6][1].numel()}", flush=True)
    if case % 10 == 9:
        print(f"{case + 1} cases, {bad} mismatches", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
