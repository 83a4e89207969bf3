"""CPU: the oracle restatement against oracle/_ref (the reference's own CPU sources compiled
by oracle/build.py).  Skipped where the prebuilt .so files and /root/reference are both absent;
the committed fixtures (test_oracle_golden.py) cover that case."""
import numpy as np
import pytest

from helpers import anchor_grid, dota_like_gt, rand_boxes
from oracle import api as O

pytestmark = pytest.mark.skipif(not O.ref_available(), reason="oracle/_ref not built")


def _eq(a, b):
    return np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize("seed", [3, 4])
def test_iou_random_bit_exact(seed):
    a = rand_boxes(400, seed, span=300.0)
    g = rand_boxes(96, seed + 50, span=300.0, amin=-np.pi, amax=np.pi)
    assert _eq(O.iou_mat(O.V1, a, g), O.ref_v1_iou_mat(a, g))
    assert _eq(O.iou_mat(O.V1, g, a, iof=True), O.ref_v1_iou_mat(g, a, True))
    assert _eq(O.iou_mat(O.V3, a, g), O.ref_v3_iou_mat(a, g))
    assert _eq(O.iou_mat(O.V3, g, a, iof=True), O.ref_v3_iou_mat(g, a, True))
    z = np.zeros((len(a), 1), np.float32)
    zg = np.zeros((len(g), 1), np.float32)
    assert _eq(O.iou_mat(O.V2, a, g), O.ref_v2_iou_mat(np.hstack([a, z]), np.hstack([g, zg])))


def test_iou_assignment_shaped_bit_exact():
    """Axis-aligned grid anchors (theta = 0 exactly: parallel / collinear edge branches) vs
    DOTA-like GT -- the shape MaxIoUAssigner feeds RBboxOverlaps2D_v1."""
    anchors = anchor_grid()[::37]  # 5309 anchors across all levels
    gt = dota_like_gt(24, 7)
    assert _eq(O.iou_mat(O.V1, gt, anchors), O.ref_v1_iou_mat(gt, anchors))
    assert _eq(O.iou_mat(O.V3, gt, anchors[:1500]), O.ref_v3_iou_mat(gt, anchors[:1500]))
    # anchors against anchors: many exactly shared edges / identical boxes
    sub = anchor_grid(256)[::5]
    assert _eq(O.iou_mat(O.V1, sub[:300], sub), O.ref_v1_iou_mat(sub[:300], sub))
    assert _eq(O.iou_mat(O.V3, sub[:120], sub[:900]), O.ref_v3_iou_mat(sub[:120], sub[:900]))


def test_iou_vec_matches_matrix_diagonal():
    a, b = rand_boxes(200, 8, span=150.0), rand_boxes(200, 9, span=150.0)
    for geom, ref in [(O.V1, O.ref_v1_iou_mat), (O.V3, O.ref_v3_iou_mat)]:
        assert _eq(O.iou_vec(geom, a, b), np.diag(ref(a, b)))
    one = a[:1]
    assert _eq(O.iou_vec(O.V1, one, b), O.ref_v1_iou_mat(one, b)[0])  # modulo broadcast


@pytest.mark.parametrize("n,span", [(300, 250.0), (1200, 500.0)])
def test_nms_keep_bit_exact(n, span):
    b = rand_boxes(n, 40 + n, span=span)
    r = np.random.default_rng(n)
    s = r.uniform(0, 1, n).astype(np.float32)
    lab = r.integers(0, 15, n).astype(np.float32)
    for thr in (0.1, 0.3):
        assert np.array_equal(O.nms(O.V1, b, s, thr, ascending=True),
                              O.ref_v1_rnms(np.hstack([b, s[:, None]]), thr))
        assert np.array_equal(O.nms(O.V3, b, s, thr), O.ref_v3_nms(b, s, thr))
        bl = np.hstack([b, lab[:, None]])
        assert np.array_equal(O.nms(O.V2, bl, s, thr, with_label=True), O.ref_v2_nms(bl, s, thr))


def test_nms_tied_scores_are_stable():
    b = np.array([[50, 50, 20, 10, 0], [500, 500, 20, 10, 0], [52, 50, 20, 10, 0]], np.float32)
    s = np.array([0.5, 0.5, 0.5], np.float32)
    assert list(O.ref_v1_rnms(np.hstack([b, s[:, None]]), 0.1)) == [0, 1]
    assert list(O.nms(O.V1, b, s, 0.1, ascending=True)) == [0, 1]
    assert list(O.nms(O.V3, b, s, 0.1)) == list(O.ref_v3_nms(b, s, 0.1)) == [0, 1]
