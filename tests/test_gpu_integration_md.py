"""INTEGRATION.md path B, executed as written: the `_hip.py` stub and three rows of its call table are extracted from
the markdown and run verbatim (exec) against the outputs recorded from the reference's own wrappers
(tests/golden/wrappers.npz, iou_random.npz).  A maintainer who pastes them gets what the reference returned."""
import os
import re

import numpy as np
import pytest
import torch

from helpers import GOLDEN

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MD = open(os.path.join(ROOT, "INTEGRATION.md")).read()
LIB = os.path.join(ROOT, "r3det-pytorch_amd", "libr3det_hip.so")


def stub_namespace():
    m = re.search(r"```python\n(# r3det/ops/_hip\.py.*?)```", MD, flags=re.S)
    assert m, "the _hip.py stub is gone from INTEGRATION.md"
    src = m.group(1)
    assert 'ctypes.CDLL("libr3det_hip.so")' in src
    src = src.replace('ctypes.CDLL("libr3det_hip.so")', f'ctypes.CDLL({LIB!r})')  # (the only edit: where the file lies)
    ns = {}
    exec(src, ns)
    return ns


def row_code(first_cell_starts_with):
    for line in MD.splitlines():
        if line.startswith("| `" + first_cell_starts_with):
            cell = line.split(" | ")[1]
            m = re.match(r"`([^`]*)`", cell)
            assert m, line
            return m.group(1).replace("; ", "\n")
    raise AssertionError(f"row {first_cell_starts_with!r} is gone from INTEGRATION.md")


def test_mat_iou_iof_row():
    g = np.load(os.path.join(GOLDEN, "iou_random.npz"))
    ns = stub_namespace()
    dev = torch.device("cuda")
    rb1, rb2 = torch.from_numpy(g["anchors"]).to(dev), torch.from_numpy(g["gts"]).to(dev)
    ns.update(torch=torch, dev=dev, rb1=rb1, rb2=rb2, n1=rb1.size(0), n2=rb2.size(0), iof=False)
    exec(row_code("rbbox_geo_cuda.mat_iou_iof(rb1, rb2, iof)"), ns)
    torch.cuda.synchronize()
    assert np.abs(ns["out"].cpu().numpy() - g["v1_iou"]).max() <= 1e-5


def test_rnms_row():
    g = np.load(os.path.join(GOLDEN, "wrappers.npz"))
    ns = stub_namespace()
    dev = torch.device("cuda")
    d6 = torch.from_numpy(g["small_dets6"]).to(dev)
    ns.update(torch=torch, dev=dev, dets6=d6, n=d6.size(0), thr=0.1)
    exec(row_code("rnms_ext.rnms(dets6, thr)"), ns)
    assert np.array_equal(ns["keep"].cpu().numpy(), g["small_rnms_keep"])


@pytest.mark.parametrize("agnostic", [0, 1])
def test_batched_rnms_row(agnostic):
    g = np.load(os.path.join(GOLDEN, "wrappers.npz"))
    ns = stub_namespace()
    dev = torch.device("cuda")
    b, s, lab = (torch.from_numpy(g[k]).to(dev) for k in ("b_boxes", "b_scores", "b_labels"))
    ns.update(torch=torch, dev=dev, bboxes=b, scores=s, inds=None if agnostic else lab.long(), n=b.size(0), nms_thr=0.1)
    exec(row_code("batched_rnms(bboxes, scores, inds, nms_thr)"), ns)
    assert np.array_equal(ns["keep"].cpu().numpy(), g[f"batched_rnms_{agnostic}_keep"])
    assert np.array_equal(ns["dets"].cpu().numpy(), g[f"batched_rnms_{agnostic}_dets"])
