"""Wrapper-level golden vectors: run the REFERENCE's own Python wrappers
(r3det/ops/rnms/rnms_wrapper.py, ops/nms_rotated/nms_rotated_wrapper.py,
ops/box_iou_rotated/box_iou_rotated_wrapper.py, core/post_processing/bbox_nms_rotated.py) on CPU
tensors in the build container and record inputs + outputs.

The wrappers import compiled extension modules that cannot be built here as torch extensions
without the CUDA halves; they are imported from where they lie with the extension names bound to
thin shims over oracle/_ref (the reference's own CPU C++ sources, compiled by oracle/build.py).
mmcv / mmdet are not installed: `mmcv.ops.nms_rotated` is only referenced by the 'mmcv' branch,
which is not exercised.  Only data is written (tests/golden/wrappers.npz).

    python tests/golden/make_golden_wrappers.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import api as O  # noqa: E402

REF = os.environ.get("R3DET_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def _mod(name, is_pkg=False, **attrs):
    m = types.ModuleType(name)
    if is_pkg:
        m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, path))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def bind_reference():
    """Import the reference wrappers with their extension modules bound to oracle/_ref."""
    def rnms(dets, thr):
        return torch.from_numpy(O.ref_v1_rnms(dets.numpy(), thr))

    def nms_rotated(b, s, thr):
        return torch.from_numpy(O.ref_v3_nms(b.contiguous().numpy(), s.contiguous().numpy(), thr))

    def overlaps(b1, b2, iou):
        return torch.from_numpy(O.ref_v3_iou_mat(b1.contiguous().numpy(), b2.contiguous().numpy(), iof=not iou))

    def ml_nms_rotated(dets, scores, labels, thr):
        d6 = torch.cat([dets, labels.unsqueeze(1).to(dets)], 1)  # at::cat({dets, labels}) promotes
        return torch.from_numpy(O.ref_v2_nms(d6.numpy(), scores.contiguous().numpy(), thr))

    _mod("r3det", True)
    _mod("r3det.ops", True)
    _mod("r3det.ops.rnms", True)
    _mod("r3det.ops.rnms.rnms_ext", rnms=rnms)
    w1 = _load("r3det.ops.rnms.rnms_wrapper", "r3det/ops/rnms/rnms_wrapper.py")
    _mod("r3det.ops.nms_rotated", True)
    _mod("r3det.ops.nms_rotated.nms_rotated_ext", nms_rotated=nms_rotated)
    w3 = _load("r3det.ops.nms_rotated.nms_rotated_wrapper", "r3det/ops/nms_rotated/nms_rotated_wrapper.py")
    _mod("r3det.ops.convex", True, convex_sort=None)
    _mod("r3det.ops.box_iou_rotated", True)
    _mod("r3det.ops.box_iou_rotated.box_iou_rotated_ext", overlaps=overlaps)
    wi = _load("r3det.ops.box_iou_rotated.box_iou_rotated_wrapper",
               "r3det/ops/box_iou_rotated/box_iou_rotated_wrapper.py")
    ops = sys.modules["r3det.ops"]
    ops.batched_rnms, ops.rnms = w1.batched_rnms, w1.rnms
    ops.obb_batched_nms, ops.obb_nms = w3.obb_batched_nms, w3.obb_nms
    ops.ml_nms_rotated = ml_nms_rotated
    ops.obb_overlaps = wi.obb_overlaps
    _mod("mmcv", True)
    _mod("mmcv.ops", nms_rotated=None)
    _mod("r3det.core", True)
    _mod("r3det.core.post_processing", True)
    pp = _load("r3det.core.post_processing.bbox_nms_rotated", "r3det/core/post_processing/bbox_nms_rotated.py")
    return types.SimpleNamespace(rnms=w1.rnms, batched_rnms=w1.batched_rnms, obb_nms=w3.obb_nms,
                                 obb_batched_nms=w3.obb_batched_nms, obb_overlaps=wi.obb_overlaps,
                                 multiclass_nms_rotated=pp.multiclass_nms_rotated)


def rand_boxes(n, seed, span=500.0, lo=8.0, hi=128.0):
    r = np.random.default_rng(seed)
    return np.stack([r.uniform(0, span, n), r.uniform(0, span, n), r.uniform(lo, hi, n),
                     r.uniform(lo, hi, n), r.uniform(-np.pi / 2, 0, n)], 1).astype(np.float32)


def main():
    assert os.path.isdir(REF)
    R = bind_reference()
    out = {}
    n, C = 1500, 15
    mb = rand_boxes(n, 91)
    ms = (np.random.default_rng(92).uniform(0, 1, (n, C + 1)) ** 12).astype(np.float32)
    out["mc_boxes"], out["mc_scores"] = mb, ms
    for ver in ("v1", "v2", "v3", "default"):
        for max_num in (50, 2000):
            cfg = dict(iou_thr=0.1) if ver == "default" else dict(type=ver, iou_thr=0.1)

            class Cfg(dict):
                __getattr__ = dict.__getitem__
            d, l = R.multiclass_nms_rotated(torch.from_numpy(mb), torch.from_numpy(ms), 0.05, Cfg(cfg), max_num)
            out[f"mc_{ver}_{max_num}_dets"] = d.numpy()
            out[f"mc_{ver}_{max_num}_labels"] = l.numpy()
            print("multiclass", ver, max_num, d.shape)
    # batched helpers
    b = rand_boxes(1200, 95, span=400.0)
    s = np.random.default_rng(96).uniform(0.05, 1, 1200).astype(np.float32)
    lab = np.random.default_rng(97).integers(0, 15, 1200)
    out["b_boxes"], out["b_scores"], out["b_labels"] = b, s, lab
    for name, fn in (("rnms", R.batched_rnms), ("obb", R.obb_batched_nms)):
        for agn in (False, True):
            d, k = fn(torch.from_numpy(b), torch.from_numpy(s), torch.from_numpy(lab), 0.1, class_agnostic=agn)
            out[f"batched_{name}_{int(agn)}_dets"], out[f"batched_{name}_{int(agn)}_keep"] = d.numpy(), k.numpy()
    # too-small handling
    small = rand_boxes(400, 98, span=200.0)
    small[::17, 2] = 5e-4
    small[5::23, 3] = 1e-4
    sc = np.random.default_rng(99).uniform(0.05, 1, 400).astype(np.float32)
    d6 = np.hstack([small, sc[:, None]])
    out["small_dets6"] = d6
    d, k = R.obb_nms(torch.from_numpy(d6), 0.1)
    out["small_obb_nms_keep"] = k.numpy()
    d, k = R.rnms(torch.from_numpy(d6), 0.1)
    out["small_rnms_keep"] = k.numpy()
    other = rand_boxes(150, 100, span=200.0)
    other[3::29, 2] = 2e-4
    out["small_other"] = other
    out["small_obb_overlaps_iou"] = R.obb_overlaps(torch.from_numpy(small), torch.from_numpy(other)).numpy()
    out["small_obb_overlaps_iof"] = R.obb_overlaps(torch.from_numpy(small), torch.from_numpy(other), mode='iof').numpy()
    out["np_obb_overlaps"] = R.obb_overlaps(small[:50], other[:40])  # numpy in -> numpy out
    np.savez_compressed(os.path.join(OUT, "wrappers.npz"), **out)
    print("done")


if __name__ == "__main__":
    main()
