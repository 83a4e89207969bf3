"""r3det_mcnms_select (threshold + ordered compaction of the (row, class) candidates, bbox_nms_rotated.py:97-114's
`nonzero` order): the one-launch form of round 5 against the two-launch form of rounds 2-4 (option nms_impl 5) and against
the torch statement of the same selection -- bit-exact (indices, scores, counts, the box maximum)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _select(boxes, scores, thr):
    from r3det import _C
    L = _C.lib()
    B, n, K1 = scores.shape
    K = K1 - 1
    dev = boxes.device
    S = n * K
    row = torch.full((B, S), -7, dtype=torch.int32, device=dev)
    lab = torch.full((B, S), -7, dtype=torch.int32, device=dev)
    sc = torch.full((B, S), -7.0, dtype=torch.float32, device=dev)
    rk = torch.full((B, S), -7, dtype=torch.int32, device=dev)
    cnt = torch.full((B,), -7, dtype=torch.int32, device=dev)
    mx = torch.full((B,), -7.0, dtype=torch.float32, device=dev)
    nb = int(L.r3det_mcnms_select_workspace_bytes(B, n))
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    _C.check(L.r3det_mcnms_select(_C.ptr(boxes), _C.ptr(scores), B, n, K, float(thr), _C.ptr(row), _C.ptr(lab), _C.ptr(sc),
                                  _C.ptr(rk), _C.ptr(cnt), _C.ptr(mx), _C.ptr(ws), nb, _C.stream()), "r3det_mcnms_select")
    torch.cuda.synchronize()
    return row, lab, sc, cnt, mx


@pytest.mark.parametrize("B,n,K,thr", [(1, 100, 15, 0.05), (2, 1024, 15, 0.3), (3, 5344, 15, 0.05), (2, 3000, 7, 0.2),
                                       (1, 2500, 80, 0.6), (2, 1500, 15, 2.0), (1, 1, 15, 0.0), (2, 4097, 3, 0.5)])
def test_select_forms_and_torch(B, n, K, thr):
    from r3det import _C
    g = torch.Generator().manual_seed(B * 1000 + n + K)
    boxes = (torch.rand(B, n, 5, generator=g) * 900 - 50).cuda()
    scores = torch.rand(B, n, K + 1, generator=g).cuda()
    scores[:, ::7] = 0.0  # rows without a candidate
    new = _select(boxes, scores, thr)
    _C.set_option("nms_impl", 5)
    try:
        old = _select(boxes, scores, thr)
    finally:
        _C.set_option("nms_impl", 0)
    for b in range(B):
        m = scores[b, :, :K] > thr
        idx = torch.nonzero(m)  # (row, class), rows ascending, classes ascending inside a row
        c = idx.size(0)
        for got in (new, old):
            row, lab, sc, cnt, mx = got
            assert int(cnt[b]) == c
            assert torch.equal(row[b, :c].long(), idx[:, 0]) and torch.equal(lab[b, :c].long(), idx[:, 1])
            assert torch.equal(sc[b, :c], scores[b, :, :K][m])
            assert int((row[b, c:] != -7).sum()) == 0  # nothing written beyond the count
            rows_with = m.any(1)
            want = boxes[b][rows_with].max() if c else torch.tensor(float("-inf"))
            assert float(mx[b]) == float(want)
