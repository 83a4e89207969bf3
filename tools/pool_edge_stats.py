"""Suppressor-edge statistics of the bench model's own NMS pool (image 0): per class candidates, suppressors per row,
rows with more than 32, greedy dependency depth.  POOL_SPREAD=0: the one-label calibration."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402
from r3det.ops import rbbox_iou  # noqa: E402

dev = torch.device("cuda", 0)
model, img = bench.build_model(dev, 100, spread=os.environ.get("POOL_SPREAD", "1") == "1")
boxes, scores = model.dense_test(img)
b, s = boxes[0], scores[0, :, :-1]
tot, over, depth_max, ecs = 0, 0, 0, []
rounds_all, und_tot = [], {}
for c in range(s.size(1)):
    idx = torch.nonzero(s[:, c] > 0.05).squeeze(1)
    if idx.numel() == 0:
        continue
    order = s[idx, c].argsort(descending=True, stable=True)
    bb = b[idx][order].contiguous()
    m = bb.size(0)
    iou = rbbox_iou(bb, bb)
    sup = torch.triu(iou > 0.1, diagonal=1)          # sup[i, j]: i (higher score) suppresses j
    ec = sup.sum(0)
    ecs.append(ec)
    tot += m
    over += int((ec > 32).sum())
    # greedy + depth
    kept = torch.zeros(m, dtype=torch.bool, device=dev)
    depth = torch.zeros(m, dtype=torch.int64, device=dev)
    supc = sup.cpu()
    keptc, depthc = kept.cpu(), depth.cpu()
    for j in range(m):
        si = torch.nonzero(supc[:, j]).squeeze(1)
        if si.numel() == 0:
            keptc[j] = True
            continue
        keptc[j] = not bool(keptc[si].any())
        depthc[j] = int(depthc[si].max()) + 1
    depth_max = max(depth_max, int(depthc.max()))
    # the round reducer's own schedule: a row is removed once a suppressor is known kept, kept once all are known removed
    state = torch.zeros(m, dtype=torch.int8)  # 0 undecided, 1 kept, 2 removed
    supf = supc.float()
    und_after = {}
    for rnd in range(1, 200):
        keptv, remv = (state == 1).float(), (state == 2).float()
        any_kept = (supf * keptv[:, None]).sum(0) > 0
        all_rem = (supf * (1 - remv)[:, None]).sum(0) == 0
        new = state.clone()
        new[(state == 0) & any_kept] = 2
        new[(state == 0) & ~any_kept & all_rem] = 1
        state = new
        und = int((state == 0).sum())
        if rnd in (4, 8, 16, 32):
            und_after[rnd] = und
        if und == 0:
            break
    rounds_all.append(rnd)
    for k_ in (4, 8, 16, 32):
        und_tot[k_] = und_tot.get(k_, 0) + und_after.get(k_, 0)
    if c < 3:
        print(f"class {c}: {m} candidates, kept {int(keptc.sum())}, suppressors per row mean {float(ec.float().mean()):.1f} max {int(ec.max())}, "
              f"rows > 32: {int((ec > 32).sum())}, dependency depth {int(depthc.max())}")
ec = torch.cat(ecs).float()
print(f"image 0: {tot} candidates; suppressors per row mean {float(ec.mean()):.1f}, median {float(ec.median()):.0f}, 90 % {float(ec.quantile(0.9)):.0f}, "
      f"max {int(ec.max())}; rows with > 32: {over} ({over / tot:.1%}); deepest dependency chain {depth_max}")
print(f"dependency rounds per class until every row is decided: max {max(rounds_all)}, mean {sum(rounds_all) / len(rounds_all):.1f}; "
      f"rows still undecided after 4 / 8 / 16 / 32 rounds (all classes): {und_tot.get(4, 0)} / {und_tot.get(8, 0)} / {und_tot.get(16, 0)} / {und_tot.get(32, 0)}")
