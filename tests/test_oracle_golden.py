"""CPU: the oracle restatement against the committed golden vectors (produced by the
reference's own CPU code, tests/golden/make_golden.py) and the SURVEY appendix-C table."""
import os

import numpy as np
import pytest

from helpers import GOLDEN
from oracle import api as O


def _load(name):
    return np.load(os.path.join(GOLDEN, name))


def _eq(a, b):
    return np.array_equal(a, b, equal_nan=True)


def test_iou_random_bit_exact():
    g = _load("iou_random.npz")
    a, gt = g["anchors"], g["gts"]
    assert _eq(O.iou_mat(O.V1, a, gt), g["v1_iou"])
    assert _eq(O.iou_mat(O.V1, a, gt, iof=True), g["v1_iof"])
    assert _eq(O.iou_mat(O.V1, gt, a), g["v1_iou_t"])
    assert _eq(O.iou_mat(O.V3, a, gt), g["v3_iou"])
    assert _eq(O.iou_mat(O.V3, a, gt, iof=True), g["v3_iof"])
    assert _eq(O.iou_mat(O.V2, a, gt), g["v2_iou"])
    da, dg = g["dense_a"], g["dense_g"]
    assert _eq(O.iou_mat(O.V1, da, dg), g["dense_v1_iou"])
    assert _eq(O.iou_mat(O.V3, da, dg), g["dense_v3_iou"])
    assert _eq(O.iou_mat(O.V2, da, dg), g["dense_v2_iou"])
    assert (g["dense_v1_iou"] > 0).mean() > 0.2  # the dense set really exercises the clipper


def test_iou_degenerate_bit_exact():
    g = _load("iou_degenerate.npz")
    d = g["boxes"]
    for key, geom, iof in [("v1_iou", O.V1, False), ("v1_iof", O.V1, True), ("v3_iou", O.V3, False),
                           ("v3_iof", O.V3, True), ("v2_iou", O.V2, False)]:
        ref = g[key]
        got = O.iou_mat(geom, d, d, iof=iof)
        ok = ref != -2.0  # -2 marks pairs where the reference overruns its 16-point scratch (UB)
        assert _eq(got[ok], ref[ok]), key
    assert (g["v1_iou"] == -2.0).sum() > 0  # the fixture does contain such pairs


def test_appendix_c_known_answers():
    """values produced by the reference CPU sources."""
    A = np.array([[50, 50, 20, 10, 0]], np.float32)
    cases = [
        ([50, 50, 20, 10, 0], [50, 50, 20, 10, 0], 1.0, 1.0, 1.0, 1.0),
        ([50, 50, 20, 10, 0], [60, 50, 20, 10, 0], 0.33333334, 0.5, 0.33333334, 0.5),
        ([50, 50, 20, 10, 0], [70, 50, 20, 10, 0], 0, 0, 0, 0),
        ([50, 50, 20, 10, 0], [70, 60, 20, 10, 0], 0, 0, 0, 0),
        ([50, 50, 40, 40, 0], [50, 50, 10, 10, -0.5], 0.0625, 0.0625, 0.0625, 0.0625),
        ([50, 50, 40, 10, 0], [50, 50, 10, 40, 0], 0.14285715, 0.25, 0.14285715, 0.25),
        ([50, 50, 20, 20, 0], [50, 50, 20, 20, -np.pi / 4], 0.70710677, 0.82842714, 0.70710677, 0.82842714),
        ([50, 50, 20, 10, 0], [500, 500, 20, 10, 0], 0, 0, 0, 0),
        ([50, 50, 20, 10, 0], [69.995, 50, 20, 10, 0], 0, 0, 1.2495e-4, 2.4986e-4),
    ]
    for a, b, v1u, v1f, v3u, v3f in cases:
        a = np.array([a], np.float32)
        b = np.array([b], np.float32)
        assert O.iou_mat(O.V1, a, b)[0, 0] == pytest.approx(v1u, rel=1e-6, abs=1e-9)
        assert O.iou_mat(O.V1, a, b, iof=True)[0, 0] == pytest.approx(v1f, rel=1e-6, abs=1e-9)
        assert O.iou_mat(O.V3, a, b)[0, 0] == pytest.approx(v3u, rel=2e-4, abs=1e-9)
        assert O.iou_mat(O.V3, a, b, iof=True)[0, 0] == pytest.approx(v3f, rel=2e-4, abs=1e-9)
    z = np.array([[50, 50, 0, 10, 0]], np.float32)
    assert np.isnan(O.iou_mat(O.V1, z, A, iof=True)[0, 0])  # 0/0: v1 has no area guard
    assert O.iou_mat(O.V3, z, A, iof=True)[0, 0] == 0


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 500, 2000])
def test_nms_keep_bit_exact(n):
    g = _load("nms.npz")
    b, s, lab = g[f"boxes_{n}"], g[f"scores_{n}"], g[f"labels_{n}"]
    for thr in (0.1, 0.5):
        tag = f"{n}_{int(thr * 100):02d}"
        assert np.array_equal(O.nms(O.V1, b, s, thr, ascending=True), g[f"v1_{tag}"])
        assert np.array_equal(O.nms(O.V3, b, s, thr), g[f"v3_{tag}"])
        bl = np.concatenate([b.reshape(-1, 5), lab.reshape(-1, 1)], 1)
        assert np.array_equal(O.nms(O.V2, bl, s, thr, with_label=True), g[f"v2_{tag}"])


def test_nms_threshold_convention():
    """CPU reference suppresses on >=, CUDA on >: they differ only at equality."""
    b = np.array([[50, 50, 20, 10, 0], [60, 50, 20, 10, 0]], np.float32)
    s = np.array([0.9, 0.8], np.float32)
    thr = float(O.iou_mat(O.V1, b[:1], b[1:])[0, 0])
    assert thr == np.float32(0.3333333432674408)
    assert list(O.nms(O.V1, b, s, thr, strict=False)) == [0]
    assert list(O.nms(O.V1, b, s, thr, strict=True)) == [0, 1]
    assert list(O.nms(O.V3, b, s, thr, strict=False)) == [0]
    assert list(O.nms(O.V3, b, s, thr, strict=True)) == [0, 1]


def test_twin_trig_is_half_ulp_accurate():
    a = np.concatenate([np.linspace(-7, 7, 20001), [0.0, -0.0, 1e-8, np.pi / 2, -np.pi / 2, 100.0]])
    a = a.astype(np.float32)
    s, c = O.sincos(a)
    s64, c64 = np.sin(a.astype(np.float64)), np.cos(a.astype(np.float64))
    # correctly rounded except for (at most) a handful of double-rounding ties
    assert (s != s64.astype(np.float32)).sum() <= 2
    assert (c != c64.astype(np.float32)).sum() <= 2
    assert s[20001] == 0 and c[20001] == 1


def test_twin_mode_stays_within_tolerance_of_reference_mode():
    """The twin (deterministic trig + device sort) is what the HIP kernels reproduce bit for
    bit; it must itself sit within the 1e-5 IoU tolerance of the reference-faithful mode."""
    g = _load("iou_random.npz")
    da, dg = g["dense_a"], g["dense_g"]
    for geom, key in [(O.V1, "dense_v1_iou"), (O.V2, "dense_v2_iou"), (O.V3, "dense_v3_iou")]:
        with O.twin():
            t = O.iou_mat(geom, da, dg)
        assert np.abs(t - g[key]).max() <= 1e-5
