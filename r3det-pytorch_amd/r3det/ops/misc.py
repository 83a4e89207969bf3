"""Exported ops of the reference outside its shipped training / inference configs
(SURVEY.md 8f rank 4): ``convex_sort`` and ``polygon_iou`` over include/r3det_hip.h."""
import torch

from .. import _C


def convex_sort(pts, masks, circular=True):
    """Graham-style index ordering (convex/convex_wrapper.py:25-27, convex_ext.convex_sort):
    pts (B, P, 2) fp32 and masks (B, P) bool on the HIP device -> int64 (B, P + circular), the
    start point (lowest valid y) first, padded with -1; non-differentiable like the reference
    (ConvexSortFunction marks its output so).  Points with equal sort keys are visited in index
    order (the reference inherits whatever torch.argsort does with ties)."""
    pts = _C.need_hip(pts.detach().contiguous(), "pts")
    if not masks.is_cuda:
        raise RuntimeError("masks must be a CUDA tensor")
    if pts.dim() != 3 or pts.size(2) != 2 or masks.shape != pts.shape[:2]:
        raise RuntimeError("pts must be (B, P, 2) and masks (B, P)")
    B, P = pts.shape[:2]
    m = masks.to(torch.uint8).contiguous()
    out = torch.full((B, P + 1 if circular else P), -1, dtype=torch.int64, device=pts.device)
    if B == 0 or P == 0:
        return out
    with torch.cuda.device(pts.device):
        ws = torch.empty(B * P, dtype=torch.int32, device=pts.device)
        _C.check(_C.lib().r3det_convex_sort(_C.ptr(pts), _C.ptr(m), B, P, int(bool(circular)), _C.ptr(ws),
                                            ws.numel() * 4, _C.ptr(out), _C.stream()), "r3det_convex_sort")
    return out


def polygon_iou(poly1, poly2):
    """IoU of two batches of 4-point polygons, (n, 8) x (m, 8) -> (n, m)
    (polygon_geo/polygon_geo.py:4-6 -> polygon_geo_cpu.polygon_iou).  The reference op is CPU-only
    (DOTA evaluation, datasets/dota1.py:681); here the pairs are clipped on the HIP device: CPU
    inputs are staged on the current device and the result comes back as a CPU tensor of the input
    dtype (arithmetic is fp32)."""
    if poly1.dim() != 2 or poly2.dim() != 2 or poly1.size(1) != 8 or poly2.size(1) != 8:
        raise RuntimeError("polygons must have shape (n, 8)")
    was_cpu = not poly1.is_cuda
    dev = poly1.device if poly1.is_cuda else torch.device('cuda', torch.cuda.current_device())
    a = poly1.to(device=dev, dtype=torch.float32).contiguous()
    b = poly2.to(device=dev, dtype=torch.float32).contiguous()
    out = torch.zeros((a.size(0), b.size(0)), dtype=torch.float32, device=dev)
    if a.size(0) and b.size(0):
        with torch.cuda.device(dev):
            _C.check(_C.lib().r3det_polygon_iou(_C.ptr(a), a.size(0), _C.ptr(b), b.size(0), _C.ptr(out),
                                                _C.stream()), "r3det_polygon_iou")
    out = out.to(poly1.dtype)
    return out.cpu() if was_cpu else out
