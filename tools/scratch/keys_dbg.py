import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
from test_gpu_fr_keys import level_inputs, _index_into, PYRAMID
from r3det.ops.feature_refine import fr_forward_levels_nhwc, tap_tables, fr_backward_index_levels
N, levels = 2, PYRAMID[:1]
feats, boxes = level_inputs(N, 8, levels, 5, "piles")
scales = [1 / s for _, s in levels]; shapes = [(hw, hw) for hw, _ in levels]
tabs = tap_tables(N, shapes, "cuda")
fr_forward_levels_nhwc(feats, boxes, scales, 1, [torch.empty_like(f) for f in feats], tabs)
nb = fr_backward_index_levels(boxes, N, 8, shapes, scales, 1, None)[1]
wa = torch.zeros(nb, dtype=torch.uint8, device="cuda"); wb = wa.clone()
_index_into(wa, boxes, None, N, shapes, scales, nhwc=False, C=8)
_index_into(wb, boxes, tabs, N, shapes, scales, nhwc=False, C=8)
a = wa.cpu().numpy().view(np.int32); b = wb.cpu().numpy().view(np.int32)
d = np.nonzero(a != b)[0]
H = W = 128; HW = H * W
ci_words = N * HW * 2
csr_bytes = ((N * HW * 8 + 255) & ~255) + N * HW * 4 * 8 + 256
csr_bytes = (csr_bytes + 255) & ~255
print("words differing", len(d), "csr words", csr_bytes // 4, "total", len(a))
slices = HW // 64
hdr0 = csr_bytes // 4
print("hdr region", hdr0, hdr0 + N * slices)
for w in d[:40]:
    reg = "cellinfo" if w < ci_words else "entries" if w < csr_bytes // 4 else "hdr" if w < hdr0 + N * slices else "rows"
    print(w, reg, int(a[w]), int(b[w]), (w - hdr0 - N * slices) if reg == "rows" else "")
cap = 32
rows0 = hdr0 + N * slices
# rows: [N][slices][cap/8][4][64][2 entries][2 words]
per_img = slices * (cap // 8) * 4 * 64 * 2 * 2
seen = set()
for w in d:
    o = w - rows0
    n = o // per_img; o2 = o % per_img
    sl = o2 // ((cap // 8) * 4 * 64 * 4); o3 = o2 % ((cap // 8) * 4 * 64 * 4)
    lane = (o3 // 4) % 64
    cell = sl * 64 + lane
    if (n, cell) in seen: continue
    seen.add((n, cell))
    ci_a = a[(n * HW + cell) * 2:(n * HW + cell) * 2 + 2]; ci_b = b[(n * HW + cell) * 2:(n * HW + cell) * 2 + 2]
    print("image", n, "cell", cell, "row", cell // W, "col", cell % W, "band", cell // 256, "hdr a/b", a[hdr0 + n * slices + sl], b[hdr0 + n * slices + sl], "cellinfo a", ci_a, "b", ci_b)
print("cellinfo nonzero words a/b:", int((a[:ci_words] != 0).sum()), int((b[:ci_words] != 0).sum()))
wb2 = torch.zeros(nb, dtype=torch.uint8, device="cuda")
_index_into(wb2, boxes, tabs, N, shapes, scales, nhwc=False, C=8)
print("TAB vs TAB equal:", torch.equal(wb, wb2))
wa2 = torch.zeros(nb, dtype=torch.uint8, device="cuda")
_index_into(wa2, boxes, None, N, shapes, scales, nhwc=False, C=8)
print("box vs box equal:", torch.equal(wa, wa2))
for C in (8, 256):
    for nhwc in (True, False):
        x = torch.zeros(nb, dtype=torch.uint8, device="cuda"); y = x.clone()
        _index_into(x, boxes, None, N, shapes, scales, nhwc=nhwc, C=C)
        _index_into(y, boxes, tabs, N, shapes, scales, nhwc=nhwc, C=C)
        print("C", C, "nhwc", nhwc, "box vs TAB differing bytes:", int((x != y).sum()))
def lst(arr, n, cell, ln):
    sl, lane = cell // 64, cell % 64
    out = []
    for r in range(ln):
        at = ((((sl * (cap >> 3) + (r >> 3)) * 4 + ((r >> 1) & 3)) * 64 + lane) * 2 + (r & 1))
        w0 = rows0 + n * per_img + at * 2
        off = int(arr[w0 + 1]) // 8   # (ascale guessed below)
        out.append((int(arr[w0]) , int(arr[w0 + 1])))
    return out
for (n, cell) in list(sorted(seen))[:2]:
    la, lb = lst(a, n, cell, 20), lst(b, n, cell, 20)
    print("cell", n, cell)
    for r, (x, y) in enumerate(zip(la, lb)):
        wa_ = np.array([x[0]], dtype=np.int32).view(np.float32)[0]; wb_ = np.array([y[0]], dtype=np.int32).view(np.float32)[0]
        print(f"  r {r:2d}  box w {wa_:.5f} off {x[1]:7d} (cell {x[1] / 8:9.1f})   TAB w {wb_:.5f} off {y[1]:7d}", "" if x == y else "  <--")
