"""Seeded synthetic inputs of the shapes the R3Det / Rotated-RetinaNet configs produce at
1024 x 1024.  Used by bench.py and the tests; no dataset, no checkpoint."""
import math

import torch

STRIDES = (8, 16, 32, 64, 128)
IMG = 1024
NUM_CLASSES = 15
# per-level candidates after nms_pre=2000 in the R3Det refine head (1 anchor / position):
# min(2000, H*W) -> 2000, 2000, 1024, 256, 64 = 5344
R3DET_POOL = 5344
RRETINA_POOL = 8576


def _gen(seed, device):
    g = torch.Generator(device='cpu')
    g.manual_seed(seed)
    return g


def rand_rboxes(n, seed, span=1024.0, lo=8.0, hi=128.0, device='cpu'):
    """cx,cy ~ U(0,span); w,h ~ U(lo,hi); theta ~ U(-pi/2, 0)  (BASELINE.md section 3)."""
    g = _gen(seed, device)
    u = torch.rand(n, 5, generator=g)
    b = torch.stack([u[:, 0] * span, u[:, 1] * span, lo + u[:, 2] * (hi - lo), lo + u[:, 3] * (hi - lo),
                     -u[:, 4] * (math.pi / 2)], 1)
    return b.to(device)


def dota_like_rboxes(n, seed, size=IMG, wmin=10.0, wmax=300.0, max_aspect=8.0, device='cpu'):
    g = _gen(seed, device)
    u = torch.rand(n, 5, generator=g)
    w = torch.exp(math.log(wmin) + u[:, 2] * (math.log(wmax) - math.log(wmin)))
    h = torch.clamp(w / torch.exp(u[:, 3] * math.log(max_aspect)), min=4.0)
    return torch.stack([u[:, 0] * size, u[:, 1] * size, w, h, -u[:, 4] * (math.pi / 2)], 1).to(device)


def anchor_grid(size=IMG, strides=STRIDES, device='cpu'):
    """RetinaNet anchors as (cx, cy, w, h, 0): octave_base_scale 4, 3 scales, ratios [1, .5, 2];
    position-major (y outer, x inner), anchor-minor (ratio outer, scale inner) -- the grid of
    core/anchor/ranchor_generator.py (fp32 corner arithmetic of mmdet's AnchorGenerator included)."""
    from .core.anchor import RAnchorGenerator
    gen = RAnchorGenerator(list(strides), [1.0, 0.5, 2.0], octave_base_scale=4, scales_per_octave=3)
    return torch.cat(gen.grid_priors([(size // s, size // s) for s in strides], device='cpu')).to(device)


def fr_level_boxes(N, H, W, stride, seed, jitter=0.1, adversarial=False, device='cpu'):
    """Boxes fed to the FR sampler, (N*H*W, 5): decoded best anchors = cell centre + N(0, jitter)
    deltas (realistic locality), or uniform-random centres (worst-case gather)."""
    g = _gen(seed, device)
    n = N * H * W
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32),
                            indexing='ij')
    b = torch.zeros(n, 5)
    if adversarial:
        b[:, 0] = (torch.rand(n, generator=g) * (W + 4) - 2) * stride
        b[:, 1] = (torch.rand(n, generator=g) * (H + 4) - 2) * stride
    else:
        b[:, 0] = (xs.reshape(-1) * stride).repeat(N) + torch.randn(n, generator=g) * jitter * 4 * stride
        b[:, 1] = (ys.reshape(-1) * stride).repeat(N) + torch.randn(n, generator=g) * jitter * 4 * stride
    b[:, 2] = (2 + 6 * torch.rand(n, generator=g)) * stride
    b[:, 3] = (2 + 6 * torch.rand(n, generator=g)) * stride
    b[:, 4] = -torch.rand(n, generator=g) * (math.pi / 2)
    return b.to(device)


def fr_pyramid(N, C, seed, size=IMG, strides=STRIDES, adversarial=False, device='cpu'):
    """(features, boxes) per level for a batch of N size x size tiles."""
    g = _gen(seed, device)
    feats, boxes = [], []
    for i, s in enumerate(strides):
        f = size // s
        feats.append(torch.randn(N, C, f, f, generator=g).to(device))
        boxes.append(fr_level_boxes(N, f, f, s, seed * 131 + i, adversarial=adversarial, device=device))
    return feats, boxes


def nms_pool(n, seed, num_classes=NUM_CLASSES, size=IMG, frac_pos=0.6, device='cpu'):
    """Pre-NMS pool of one image as multiclass_nms_rotated receives it: multi_bboxes (n,5) and
    multi_scores (n, C+1).  About frac_pos of the boxes have one class above score_thr=0.05
    (a trained detector's sparsity; random-init weights would give zero candidates,
    SURVEY.md 7.4-6), clustered so that NMS at thr 0.1 removes most duplicates."""
    g = _gen(seed, device)
    n_obj = max(1, n // 12)
    objs = dota_like_rboxes(n_obj, seed + 1, size=size, wmax=150.0, max_aspect=4.0)
    obj_cls = torch.randint(0, num_classes, (n_obj,), generator=g)
    which = torch.randint(0, n_obj, (n,), generator=g)
    b = objs[which].clone()
    b[:, 0:2] += torch.randn(n, 2, generator=g) * (0.15 * b[:, 2:4].min(1, keepdim=True)[0])
    b[:, 2:4] *= torch.exp(torch.randn(n, 2, generator=g) * 0.1)
    b[:, 4] += torch.randn(n, generator=g) * 0.05
    scores = torch.rand(n, num_classes + 1, generator=g) * 0.04          # below the 0.05 threshold
    pos = torch.rand(n, generator=g) < frac_pos
    top = 0.05 + 0.95 * torch.rand(n, generator=g)
    idx = torch.nonzero(pos).squeeze(1)
    scores[idx, obj_cls[which[idx]]] = top[idx]
    scores[:, -1] = 0.0
    return b.to(device), scores.to(device)
