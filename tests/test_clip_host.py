"""The straight-line v1 clip of the drains (csrc/r3_clip.h), compiled for the HOST and checked against the oracle.

`v1_clip_fast` is __host__ __device__ and uses IEEE + - * / only, so the host build computes what the gfx950 build
computes.  Every pair must either be FLAGGED (the kernels then run the exact branchy form on it) or be bit-identical to
the oracle's restatement of rbbox_geo_kernel.cu:88-268 in twin mode.  No GPU needed."""
import ctypes
import os
import shutil
import subprocess

import numpy as np
import pytest

from helpers import anchor_grid, dota_like_gt, rand_boxes
from oracle import api as O

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "r3det-pytorch_amd", "csrc")
F = ctypes.POINTER(ctypes.c_float)
U8 = ctypes.POINTER(ctypes.c_uint8)


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    out = str(tmp_path_factory.mktemp("clip") / "libclip_harness.so")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-shared",
                    "-I" + CSRC, os.path.join(HERE, "native", "clip_harness.hip"), "-o", out], check=True)
    L = ctypes.CDLL(out)
    L.clip_fast_batch.argtypes = [F, F, ctypes.c_int, ctypes.c_int, F, U8]
    L.hull_clip_fast_batch.argtypes = [F, F, ctypes.c_int, ctypes.c_int, ctypes.c_int, F, U8]
    return L


def records(b):
    """make_record<1> (csrc/r3_geom.h) in numpy float32: vertices + w * h, sine / cosine from the twin routine."""
    b = np.ascontiguousarray(b, np.float32)
    x, y, w, h, a = [b[:, k] for k in range(5)]
    s, c = O.sincos(a)
    two = np.float32(2)
    with np.errstate(all="ignore"):
        w_2, h_2 = w / two, h / two
        wx, wy = c * w_2, s * w_2
        hx, hy = -s * h_2, c * h_2
        r = np.empty((b.shape[0], 9), np.float32)
        r[:, 0], r[:, 1] = x + wx + hx, y + wy + hy
        r[:, 2], r[:, 3] = x - wx + hx, y - wy + hy
        r[:, 4], r[:, 5] = x - wx - hx, y - wy - hy
        r[:, 6], r[:, 7] = x + wx - hx, y + wy - hy
        r[:, 8] = w * h
    return r


def run(L, b1, b2, iof=False):
    n = b1.shape[0]
    ra, rb = records(b1), records(b2)
    out = np.empty(n, np.float32)
    redo = np.empty(n, np.uint8)
    L.clip_fast_batch(ra.ctypes.data_as(F), rb.ctypes.data_as(F), n, int(iof), out.ctypes.data_as(F),
                      redo.ctypes.data_as(U8))
    with O.twin():
        want = O.iou_vec(O.V1, b1, b2, iof)
    ok = redo == 0
    same = (out.view(np.uint32) == want.view(np.uint32)) | (np.isnan(out) & np.isnan(want))
    assert np.all(same[ok]), f"{np.sum(~same & ok)} unflagged pairs differ from the oracle"
    return redo.mean(), want


def near_pairs(n, seed, span=300):
    r = np.random.default_rng(seed)
    a = rand_boxes(n, seed, span=span, lo=8, hi=128)
    b = a.copy()
    b[:, 0] += r.normal(0, 30, n)
    b[:, 1] += r.normal(0, 30, n)
    b[:, 2] *= np.exp(r.normal(0, .4, n))
    b[:, 3] *= np.exp(r.normal(0, .4, n))
    b[:, 4] = r.uniform(-np.pi / 2, 0, n)
    return a, b.astype(np.float32)


@pytest.mark.parametrize("iof", [False, True])
def test_overlapping_random_pairs_bit_exact_and_rarely_flagged(harness, iof):
    a, b = near_pairs(400000, 1)
    frac, want = run(harness, a, b, iof)
    assert np.mean(want > 0) > 0.5   # the sample really is overlapping pairs
    assert frac < 2e-3               # general position is the rule (measured 1.8e-4)


def test_assignment_shaped_pairs(harness):
    anc, gt = anchor_grid(), dota_like_gt(128, 3)
    r = np.random.default_rng(5)
    gi, ai = r.integers(0, 128, 1500000), r.integers(0, anc.shape[0], 1500000)
    d = np.hypot(anc[ai, 0] - gt[gi, 0], anc[ai, 1] - gt[gi, 1])
    m = d < (np.hypot(anc[ai, 2], anc[ai, 3]) + np.hypot(gt[gi, 2], gt[gi, 3])) / 2
    for first, second in ((gt[gi[m]], anc[ai[m]]), (anc[ai[m]], gt[gi[m]])):
        frac, want = run(harness, first, second)
        assert frac < 5e-3 and np.mean(want > 0) > 0.3


def test_degenerate_families_are_flagged_or_exact(harness):
    r = np.random.default_rng(11)
    n = 100000
    ia = np.stack([r.integers(0, 64, n), r.integers(0, 64, n), r.integers(1, 32, n), r.integers(1, 32, n),
                   np.zeros(n)], 1).astype(np.float32)
    ib = np.stack([r.integers(0, 64, n), r.integers(0, 64, n), r.integers(1, 32, n), r.integers(1, 32, n),
                   np.zeros(n)], 1).astype(np.float32)
    run(harness, ia, ib)                     # integer axis-aligned: shared edges, touching corners
    ib2 = ib.copy()
    ib2[:, 4] = -np.pi / 2
    run(harness, ia, ib2)
    a, b = near_pairs(100000, 7)
    frac, _ = run(harness, a, a.copy())      # identical boxes: every vertex coincides
    assert frac == 1.0
    for eps in (1e-3, 1e-2, 1e-1):
        c = a.copy()
        c[:, :4] += r.normal(0, eps, (n, 4)).astype(np.float32)
        c[:, 4] += r.normal(0, eps * 1e-2, n).astype(np.float32)
        run(harness, a, c)                   # near-duplicates: the 1e-2 de-dup merges candidates
    for sc in (1e-3, 1e-6, 1e4, 1e8, 1e14, 1e20, 1e-20):
        aa, bb = a.copy(), b.copy()
        aa[:, :4] *= sc
        bb[:, :4] *= sc
        run(harness, aa, bb)
    aa, bb = a.copy(), b.copy()
    aa[:, :2] += 1e5
    bb[:, :2] += 1e5
    run(harness, aa, bb)                     # far from the origin: coordinates lose bits, differences do not
    aa, bb = a.copy(), b.copy()
    aa[:, 3], bb[:, 2] = 0.01, 0.05
    run(harness, aa, bb)                     # thin boxes
    aa[:, 3] = 0
    frac, _ = run(harness, aa, bb)           # zero height: an edge of length 0
    assert frac == 1.0
    aa = a.copy()
    aa[:, 2] *= -1
    run(harness, aa, b)                      # negative width: clockwise vertices
    aa = a.copy()
    aa[::7, 0], aa[::11, 4], aa[::13, 2] = np.nan, np.inf, np.inf
    frac, _ = run(harness, aa, b)
    assert frac >= 1 / 7                     # non-finite boxes never take the straight-line form


# ------------------------------------------------------------------------------------------------ hull geometry (v2 / v3)
def hull_records(b):
    """make_record<2 | 3> (csrc/r3_geom.h): centre, the four half products, w * h."""
    b = np.ascontiguousarray(b, np.float32)
    x, y, w, h, a = [b[:, k] for k in range(5)]
    s, c = O.sincos(a)
    half = np.float32(0.5)
    with np.errstate(all="ignore"):
        c2, s2 = c * half, s * half
        r = np.empty((b.shape[0], 7), np.float32)
        r[:, 0], r[:, 1] = x, y
        r[:, 2], r[:, 3], r[:, 4], r[:, 5] = s2 * h, c2 * w, c2 * h, s2 * w
        r[:, 6] = w * h
    return r


def run_hull(L, geom, b1, b2, iof=False):
    n = b1.shape[0]
    ra, rb = hull_records(b1), hull_records(b2)
    out = np.empty(n, np.float32)
    redo = np.empty(n, np.uint8)
    L.hull_clip_fast_batch(ra.ctypes.data_as(F), rb.ctypes.data_as(F), n, int(geom == O.V2), int(not iof),
                           out.ctypes.data_as(F), redo.ctypes.data_as(U8))
    with O.twin():
        want = O.iou_vec(geom, b1, b2, iof)
    ok = redo == 0
    same = (out.view(np.uint32) == want.view(np.uint32)) | (np.isnan(out) & np.isnan(want))
    assert np.all(same[ok]), f"{np.sum(~same & ok)} unflagged pairs differ from the oracle"
    return redo.mean(), want


@pytest.mark.parametrize("geom", [O.V3, O.V2])
def test_hull_clip_random_and_assignment_shaped(harness, geom):
    a, b = near_pairs(300000, 1)
    frac, want = run_hull(harness, geom, a, b)
    assert np.mean(want > 0) > 0.5 and frac < 2e-3
    if geom == O.V3:
        run_hull(harness, geom, a, b, iof=True)
    anc, gt = anchor_grid(), dota_like_gt(128, 3)
    r = np.random.default_rng(5)
    gi, ai = r.integers(0, 128, 800000), r.integers(0, anc.shape[0], 800000)
    d = np.hypot(anc[ai, 0] - gt[gi, 0], anc[ai, 1] - gt[gi, 1])
    m = d < (np.hypot(anc[ai, 2], anc[ai, 3]) + np.hypot(gt[gi, 2], gt[gi, 3])) / 2
    frac, want = run_hull(harness, geom, gt[gi[m]], anc[ai[m]])
    assert frac < 5e-3 and np.mean(want > 0) > 0.3


@pytest.mark.parametrize("geom", [O.V3, O.V2])
def test_hull_clip_degenerate_families(harness, geom):
    r = np.random.default_rng(11)
    n = 100000
    ia = np.stack([r.integers(0, 64, n), r.integers(0, 64, n), r.integers(1, 32, n), r.integers(1, 32, n),
                   np.zeros(n)], 1).astype(np.float32)
    ib = np.stack([r.integers(0, 64, n), r.integers(0, 64, n), r.integers(1, 32, n), r.integers(1, 32, n),
                   np.zeros(n)], 1).astype(np.float32)
    run_hull(harness, geom, ia, ib)
    a, b = near_pairs(100000, 7)
    frac, _ = run_hull(harness, geom, a, a.copy())
    assert frac == 1.0                       # identical boxes: every numerator is zero
    for eps in (1e-3, 1e-1):
        c = a.copy()
        c[:, :4] += r.normal(0, eps, (n, 4)).astype(np.float32)
        run_hull(harness, geom, a, c)
    for sc in (1e-3, 1e-6, 1e4, 1e8, 1e14, 1e-20):
        aa, bb = a.copy(), b.copy()
        aa[:, :4] *= sc
        bb[:, :4] *= sc
        run_hull(harness, geom, aa, bb)
    aa, bb = a.copy(), b.copy()
    aa[:, :2] += 1e5
    bb[:, :2] += 1e5
    run_hull(harness, geom, aa, bb)
    aa, bb = a.copy(), b.copy()
    aa[:, 3], bb[:, 2] = 0.01, 0.05
    run_hull(harness, geom, aa, bb)
    aa[:, 3] = 0
    run_hull(harness, geom, aa, bb)          # zero height: area < 1e-14 => 0
    aa = a.copy()
    aa[:, 2] *= -1
    run_hull(harness, geom, aa, b)
    aa = a.copy()
    aa[::7, 0], aa[::11, 4], aa[::13, 2] = np.nan, np.inf, np.inf
    run_hull(harness, geom, aa, b)
