"""How many suppressor edges of the synthetic bench model's pre-NMS pool join boxes of DIFFERENT labels once the
reference's class offsets (label * (max coordinate + 1), bbox_nms_rotated.py) are applied -- the case in which the
batched reducer cannot treat the labels as independent problems."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")): sys.path.insert(0, p)
import torch
import bench
from r3det.ops import rbbox_iou
dev = torch.device("cuda", 0); torch.cuda.set_device(dev); torch.backends.cudnn.benchmark = True
model, img = bench.build_model(dev, 100)
with torch.no_grad():
    boxes, scores = model.dense_test(img)
for i in range(boxes.size(0)):
    sc = scores[i, :, :-1]
    rows, labs = torch.nonzero(sc > 0.05, as_tuple=True)
    b = boxes[i][rows].clone()
    off = labs.to(b) * (boxes[i].max() + 1)
    b[:, 0] += off; b[:, 1] += off
    n = b.size(0)
    iou = rbbox_iou(b, b)
    cross = (iou > 0.1) & (labs[:, None] != labs[None, :])
    same = (iou > 0.1) & (labs[:, None] == labs[None, :])
    same.fill_diagonal_(False)
    rows_x = cross.any(0).sum().item()
    print(f"image {i}: {n} candidates, edges same-label {int(same.sum().item()) // 2}, cross-label {int(cross.sum().item()) // 2}, rows with a cross-label partner {rows_x}, w max {b[:,2].max().item():.0f} h max {b[:,3].max().item():.0f} coordinate max {boxes[i].max().item():.0f}")
    print("   candidates per label:", torch.bincount(labs, minlength=15).tolist())
    # suppressor lists in score order (per label), rows beyond the 32-entry list, and the dependency rounds greedy needs
    if i == 0:
        import numpy as np
        sco = sc[rows, labs]
        order = torch.argsort(sco, descending=True)
        M = (same[order][:, order]).cpu().numpy()          # M[a, b]: a and b overlap (same label), positions in score order
        n_ = M.shape[0]
        sup = np.triu(M, 1)                                 # sup[a, b] (a < b): a is a suppressor of b
        cnt = sup.sum(0)
        state = np.zeros(n_, np.int8)                       # 0 undecided, 1 kept, 2 removed
        state[cnt == 0] = 1
        rounds = 0
        while (state == 0).any():
            kept, rem = state == 1, state == 2
            anyK = (sup & kept[:, None]).any(0)
            allR = ((~sup) | rem[:, None]).all(0)
            new = state.copy()
            new[(state == 0) & anyK] = 2
            new[(state == 0) & ~anyK & allR] = 1
            state = new
            rounds += 1
        print(f"   image 0: suppressors per row mean {cnt.mean():.1f} max {cnt.max()}, rows with more than 32: {(cnt > 32).sum()}, more than 8: {(cnt > 8).sum()}; dependency rounds {rounds}; kept {(state == 1).sum()}")
    if i == 0 and os.environ.get("DUMP_POOL"):
        import numpy as np
        arr = torch.cat([b.new_tensor([float(n)]), boxes[i][rows].reshape(-1), sc[rows, labs], labs.to(b)]).cpu().numpy().astype(np.float32)
        arr.tofile(os.environ["DUMP_POOL"])  # n | boxes (n, 5) | scores | labels: input of tools/probes/nms_reduce_probe
