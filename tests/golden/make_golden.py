"""Generate the golden fixtures in tests/golden/ from the REFERENCE's own CPU code.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

The expected outputs come from oracle/_ref/*.so, i.e. the reference's CPU sources compiled
where they lie (oracle/build.py).  Only data (inputs + expected outputs) is written here.

Fixtures
--------
iou_random.npz      config 1 of BASELINE.json: 1000 anchors (seed 0) x 128 GT (seed 1), v1 / v3 /
                    v2-stand-in, IoU and IoF, plus a dense set where ~1/3 of pairs overlap.
iou_degenerate.npz  identical / touching / nested / zero-area / tiny / sliver boxes
                     for v1, v3, v2.
nms.npz             keep lists of rnms_cpu (v1), nms_rotated_cpu (v3) and the ml header (v2) at
                    n in {0,1,63,64,65,500,2000,8576}, thr in {0.1,0.5}; knife-edge cases (any
                    pair IoU within 1e-4 of thr among boxes whose decision matters) are recorded
                    so GPU tests can tell ">=" (CPU) from ">" (CUDA) apart.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import api as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def rand_boxes(n, seed, span=1024.0, lo=8.0, hi=128.0):
    """BASELINE.md section 3 distribution: cx,cy~U(0,span), w,h~U(lo,hi), theta~U(-pi/2,0)."""
    r = np.random.default_rng(seed)
    return np.stack([r.uniform(0, span, n), r.uniform(0, span, n), r.uniform(lo, hi, n),
                     r.uniform(lo, hi, n), r.uniform(-np.pi / 2, 0, n)], 1).astype(np.float32)


DEGENERATE = np.array([
    [50, 50, 20, 10, 0], [50, 50, 20, 10, -0.3], [50, 50, 10, 20, -np.pi / 2], [60, 50, 20, 10, 0],
    [70, 50, 20, 10, 0], [70, 60, 20, 10, 0], [50, 50, 40, 40, 0], [50, 50, 10, 10, -0.5],
    [50, 50, 40, 10, 0], [50, 50, 10, 40, 0], [50, 50, 20, 20, 0], [50, 50, 20, 20, -np.pi / 4],
    [50, 50, 0, 10, 0], [50, 50, 5e-4, 5e-4, 0], [500, 500, 20, 10, 0], [69.995, 50, 20, 10, 0],
    [50, 50, 20, 10, np.pi / 2], [50, 50, 20, 10, np.pi], [50, 55, 20, 10, 0], [55, 50, 20, 10, 0],
    [50, 50, 20, 10, 1e-4], [50, 50, 20, 10, -1e-4], [40, 50, 20, 10, 0], [50, 40, 20, 10, 0],
    [50, 60, 20, 10, 0], [50, 50, 1e-8, 1e-8, 0], [50, 50, -20, 10, 0], [50, 50, 1e4, 1e4, 0.3],
    [1e4, 1e4, 30, 12, -1.0], [1e4 + 5, 1e4 + 3, 30, 12, -1.1], [50, 50, 20, 10, 7.0],
    [50, 50, 20, 10, -100.0], [0, 0, 1, 1, 0], [0.25, 0.25, 1, 1, -0.7],
], np.float32)


def with_col(b, col):
    return np.concatenate([b, np.asarray(col, np.float32).reshape(-1, 1)], 1)


def main():
    assert O._build.ref_available(), "needs /root/reference"
    O._build.build_ref()

    # ---------------- IoU, random ------------------------------------------------
    a, g = rand_boxes(1000, 0), rand_boxes(128, 1)
    da, dg = rand_boxes(300, 10, span=200.0), rand_boxes(150, 20, span=200.0)
    z = lambda n: np.zeros(n, np.float32)
    np.savez_compressed(
        os.path.join(OUT, "iou_random.npz"),
        anchors=a, gts=g, dense_a=da, dense_g=dg,
        v1_iou=O.ref_v1_iou_mat(a, g), v1_iof=O.ref_v1_iou_mat(a, g, True),
        v1_iou_t=O.ref_v1_iou_mat(g, a),
        v3_iou=O.ref_v3_iou_mat(a, g), v3_iof=O.ref_v3_iou_mat(a, g, True),
        v2_iou=O.ref_v2_iou_mat(with_col(a, z(1000)), with_col(g, z(128))),
        dense_v1_iou=O.ref_v1_iou_mat(da, dg), dense_v3_iou=O.ref_v3_iou_mat(da, dg),
        dense_v2_iou=O.ref_v2_iou_mat(with_col(da, z(300)), with_col(dg, z(150))))

    # ---------------- IoU, degenerate --------------------------------------------
    d = DEGENERATE
    lab = (np.arange(len(d)) % 2).astype(np.float32)
    np.savez_compressed(
        os.path.join(OUT, "iou_degenerate.npz"), boxes=d, labels=lab,
        v1_iou=O.ref_v1_iou_mat(d, d), v1_iof=O.ref_v1_iou_mat(d, d, True),   # -2 == reference UB
        v3_iou=O.ref_v3_iou_mat(d, d), v3_iof=O.ref_v3_iou_mat(d, d, True),
        v2_iou=O.ref_v2_iou_mat(with_col(d, z(len(d))), with_col(d, z(len(d)))),
        v2_iou_labelled=O.ref_v2_iou_mat(with_col(d, lab), with_col(d, lab)))

    # ---------------- NMS ---------------------------------------------------------
    out = {}
    for n in [0, 1, 63, 64, 65, 500, 2000, 8576]:
        span = {8576: 1500.0, 2000: 700.0}.get(n, 500.0)
        b = rand_boxes(n, 100 + n, span=span)
        r = np.random.default_rng(200 + n)
        s = r.uniform(0.05, 1, n).astype(np.float32)
        lab = r.integers(0, 15, n).astype(np.float32)
        out[f"boxes_{n}"], out[f"scores_{n}"], out[f"labels_{n}"] = b, s, lab
        for thr in (0.1, 0.5):
            tag = f"{n}_{int(thr * 100):02d}"
            if n == 0:
                e = np.zeros(0, np.int64)
                out[f"v1_{tag}"] = out[f"v3_{tag}"] = out[f"v2_{tag}"] = e
                continue
            out[f"v1_{tag}"] = O.ref_v1_rnms(with_col(b, s), thr)
            out[f"v3_{tag}"] = O.ref_v3_nms(b, s, thr)
            out[f"v2_{tag}"] = O.ref_v2_nms(with_col(b, lab), s, thr)  # (n = 8576: ~40 s per threshold on one core)
            print("nms", n, thr, len(out[f"v1_{tag}"]), len(out[f"v3_{tag}"]), flush=True)
    np.savez_compressed(os.path.join(OUT, "nms.npz"), **out)
    print("done")


if __name__ == "__main__":
    main()
