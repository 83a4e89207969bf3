"""GPU parity at the WRAPPER level: r3det.ops / r3det.core mirrors vs outputs recorded from the
reference's own Python wrappers running on its own CPU extension code
(tests/golden/make_golden_wrappers.py -> tests/golden/wrappers.npz).

Keep indices / labels: exact.  Detections are gathers of the inputs: exact.  IoU values: <= 1e-5
(the reference CPU code uses libm trig and the host branch of the hull sort)."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(GOLDEN, "wrappers.npz"))


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("ver", ["v1", "v2", "v3", "default"])
@pytest.mark.parametrize("max_num", [50, 2000])
def test_multiclass_nms_rotated(g, ver, max_num):
    from r3det.core.post_processing import multiclass_nms_rotated
    cfg = dict(iou_thr=0.1) if ver == "default" else dict(type=ver, iou_thr=0.1)
    dets, labels = multiclass_nms_rotated(dev(g["mc_boxes"]), dev(g["mc_scores"]), 0.05, cfg, max_num)
    assert np.array_equal(labels.cpu().numpy(), g[f"mc_{ver}_{max_num}_labels"])
    assert np.array_equal(dets.cpu().numpy(), g[f"mc_{ver}_{max_num}_dets"])


@pytest.mark.parametrize("agnostic", [False, True])
def test_batched_helpers(g, agnostic):
    from r3det.ops import batched_rnms, obb_batched_nms
    b, s, lab = dev(g["b_boxes"]), dev(g["b_scores"]), dev(g["b_labels"])
    for name, fn in (("rnms", batched_rnms), ("obb", obb_batched_nms)):
        d, k = fn(b, s, lab, 0.1, class_agnostic=agnostic)
        assert np.array_equal(k.cpu().numpy(), g[f"batched_{name}_{int(agnostic)}_keep"]), name
        assert np.array_equal(d.cpu().numpy(), g[f"batched_{name}_{int(agnostic)}_dets"]), name


def test_too_small_boxes(g):
    from r3det.ops import obb_nms, obb_overlaps, rnms
    d6 = g["small_dets6"]
    assert np.array_equal(obb_nms(dev(d6), 0.1)[1].cpu().numpy(), g["small_obb_nms_keep"])
    assert np.array_equal(rnms(dev(d6), 0.1)[1].cpu().numpy(), g["small_rnms_keep"])
    small, other = d6[:, :5], g["small_other"]
    for mode in ("iou", "iof"):
        got = obb_overlaps(dev(small), dev(other), mode=mode).cpu().numpy()
        ref = g[f"small_obb_overlaps_{mode}"]
        assert np.abs(got - ref).max() <= 1e-5
        assert np.array_equal(got == 0, ref == 0) or np.abs(got - ref)[(got == 0) != (ref == 0)].max() <= 1e-5
        # rows / columns of boxes thinner than 1e-3 are exactly zero in both
        assert (got[::17] == 0).all() and (got[:, 3::29] == 0).all()
    out = obb_overlaps(small[:50], other[:40], device_id=0)
    assert isinstance(out, np.ndarray) and np.abs(out - g["np_obb_overlaps"]).max() <= 1e-5
