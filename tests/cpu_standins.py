"""TEST-ONLY CPU stand-ins for the three HIP kernels the training step calls, so that the host glue
around them (targets, losses, detector wiring, DistributedDataParallel) can run in the CPU suite:

  rbbox_iou (assignment)  -> the oracle's v1 IoU matrix (oracle/, bit-identical to the reference CPU code)
  feature_refine          -> the vectorised torch restatement of tests/test_fr_oracle.py (autograd gives
                             the backward)
  r3det_filter_bboxes     -> not needed: the heads' own ``*_torch`` forms run on CPU tensors

The product has no CPU path (tests/test_abi.py::test_no_cpu_fallback); nothing here is importable
from the package."""
import contextlib

import numpy as np
import torch

from oracle import api as O
from test_fr_oracle import torch_fr


def _rbbox_iou(rb1, rb2, vec=False, iof=False):
    assert not vec
    out = O.iou_mat(O.V1, rb1.detach().numpy().astype(np.float32), rb2.detach().numpy().astype(np.float32), iof=iof)
    return torch.from_numpy(out)


def _feature_refine(features, best_rbboxes, spatial_scale, points=1, table=None):
    return torch_fr(features, best_rbboxes, spatial_scale, points)


def _feature_refine_levels(features, best_rbboxes, spatial_scales, points=1):
    return [torch_fr(f, b, s, points) for f, b, s in zip(features, best_rbboxes, spatial_scales)]


@contextlib.contextmanager
def cpu_kernels(twin=False):
    """``twin``: the oracle evaluates the kernels' own deterministic sincos instead of libm (bit-identical
    IoU to the HIP path: needed when exact ties between anchors decide the assignment)."""
    import r3det.core.bbox.iou_calculators.rotate_iou2d_calculator as calc
    import r3det.ops.feature_refine as frm
    saved = calc.rbbox_iou, frm.feature_refine, frm.feature_refine_levels, frm.feature_refine_module_levels
    calc.rbbox_iou, frm.feature_refine, frm.feature_refine_levels = _rbbox_iou, _feature_refine, _feature_refine_levels
    # (round 5: the module's tail as one node -- on the CPU the same three steps in plain torch)
    frm.feature_refine_module_levels = lambda a, b, x, boxes, scales, points=1: [
        xi + o for xi, o in zip(x, _feature_refine_levels([ai + bi for ai, bi in zip(a, b)], boxes, scales, points))]
    try:
        with (O.twin() if twin else contextlib.nullcontext()):
            yield
    finally:
        calc.rbbox_iou, frm.feature_refine, frm.feature_refine_levels, frm.feature_refine_module_levels = saved
