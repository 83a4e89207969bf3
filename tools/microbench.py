"""Op-level micro-benchmarks with kernel-variant A/B (interleaved rounds, one process).

    python tools/microbench.py [iou] [fr] [nms]

Prints one line per (op, shape, variant): median / min microseconds and the achieved
algorithmic GB/s (SURVEY.md 8d byte counts).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402

from r3det import _C, synthetic as syn  # noqa: E402


def time_variants(variants, rounds=7, reps=10):
    """variants: {name: callable}. Interleaved rounds; returns {name: (median_us, min_us)}."""
    res = {k: [] for k in variants}
    for fn in variants.values():
        fn()
    torch.cuda.synchronize()
    for _ in range(rounds):
        for name, fn in variants.items():
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(reps):
                fn()
            e.record()
            torch.cuda.synchronize()
            res[name].append(s.elapsed_time(e) * 1e3 / reps)
    out = {}
    for k, v in res.items():
        v = sorted(v)
        out[k] = (v[len(v) // 2], v[0])
    return out


def bench_iou():
    from r3det.ops import rbbox_iou
    from r3det.ops.mmcv_ops import box_iou_rotated
    from r3det.ops.iou import box_iou_rotated_v3
    dev = torch.device("cuda")
    anchors = syn.anchor_grid(device=dev)
    refine = syn.fr_level_boxes(1, 1, 21824, 8, 3, device=dev)  # 21 824 "rois"
    shapes = [("128x196416 assign", syn.dota_like_rboxes(128, 5, device=dev), anchors),
              ("512x196416 assign", syn.dota_like_rboxes(512, 6, device=dev), anchors),
              ("128x21824 refine", syn.dota_like_rboxes(128, 7, device=dev), syn.dota_like_rboxes(21824, 8, device=dev)),
              ("1000x128 config1", syn.rand_rboxes(1000, 0, device=dev), syn.rand_rboxes(128, 1, device=dev)),
              ("2000x2000 dense", syn.rand_rboxes(2000, 2, span=300., device=dev), syn.rand_rboxes(2000, 3, span=300., device=dev))]
    for name, a, b in shapes:
        m, n = a.size(0), b.size(0)
        alg = 4 * m * n + 20 * (m + n)

        def mk(impl, fn):
            def run():
                _C.set_option("iou_impl", impl)
                fn(a, b)
            return run
        var = {"v1 auto": mk(0, rbbox_iou), "v1 fused": mk(4, rbbox_iou), "v1 compact": mk(2, rbbox_iou),
               "v1 simple": mk(1, rbbox_iou), "v3 auto": mk(0, box_iou_rotated_v3), "v2 auto": mk(0, box_iou_rotated)}
        if m * n > 3e7:
            var.pop("v1 simple")
        for k, (med, mn) in time_variants(var).items():
            print(f"iou  {name:20s} {k:12s} med {med:9.1f} us  min {mn:9.1f} us  {m * n / med:9.1f} Mpairs/s  "
                  f"{alg / med / 1e3:8.1f} GB/s", flush=True)
    _C.set_option("iou_impl", 0)


def bench_assign():
    """MaxIoU assignment on the training-step shapes: fused (no matrix) vs dense (IoU matrix +
    torch reductions + the per-gt loop)."""
    from r3det.core.bbox.assigners import MaxIoUAssigner
    dev = torch.device("cuda")
    anchors = syn.anchor_grid(device=dev)
    for k in (32, 128):
        gts = syn.dota_like_rboxes(k, 5, device=dev)
        asg = MaxIoUAssigner(0.5, 0.4, 0., iou_calculator=dict(type='RBboxOverlaps2D_v1'))
        var = {"fused": lambda: asg.assign(anchors, gts),
               "dense": lambda: asg.assign_wrt_overlaps(asg.iou_calculator(gts, anchors)),
               "iou matrix only": lambda: asg.iou_calculator(gts, anchors)}
        for name, (med, mn) in time_variants(var, rounds=5, reps=5).items():
            print(f"assign {k}x{anchors.size(0)} {name:16s} med {med:9.1f} us  min {mn:9.1f} us", flush=True)


def bench_fr():
    from r3det.ops.feature_refine import fr_backward, fr_forward
    dev = torch.device("cuda")
    for N in (4,):
        feats, boxes = syn.fr_pyramid(N, 256, 9, device=dev)
        advf, advb = syn.fr_pyramid(N, 256, 9, adversarial=True, device=dev)
        for lvl, s in enumerate(syn.STRIDES):
            f, b = feats[lvl], boxes[lvl]
            o = torch.empty_like(f)
            g = torch.empty_like(f)
            alg = 8 * f.numel() + 20 * b.size(0)

            def mk(impl, bb, bwd=False):
                def run():
                    _C.set_option("fr_impl", impl)
                    if bwd:
                        fr_backward(f, bb, 1.0 / s, 1, g, overwrite=True)
                    else:
                        fr_forward(f, bb, 1.0 / s, 1, o)
                return run
            var = {"fwd cell": mk(10, b), "fwd plane": mk(2, b), "fwd generic": mk(1, b),
                   "fwd auto adv": mk(0, advb[lvl]),
                   "bwd packed": mk(0, b, True), "bwd cell": mk(10, b, True), "bwd plane": mk(2, b, True), "bwd generic": mk(1, b, True)}
            for k, (med, mn) in time_variants(var).items():
                print(f"fr   N={N} level{lvl} {tuple(f.shape)} {k:14s} med {med:8.1f} us  min {mn:8.1f} us  "
                      f"{alg / med / 1e3:8.1f} GB/s", flush=True)
    _C.set_option("fr_impl", 0)


def bench_nms():
    from r3det.ops import batched_rnms, ml_nms_rotated, obb_batched_nms
    dev = torch.device("cuda")
    for n in (2000, 5344, 8576, 32768):
        mb, ms = syn.nms_pool(n * 10 // 6 + 64, 77 + n, device=dev)
        sc, lab = ms[:, :-1].max(1)
        idx = torch.nonzero(sc > 0.05).squeeze(1)[:n]
        b, s, l = mb[idx].contiguous(), sc[idx].contiguous(), lab[idx].contiguous()
        nn = b.size(0)
        cb = (nn + 63) // 64
        alg = 24 * nn + 8 * nn * cb + 8 * nn
        var = {"v1 batched_rnms": lambda: batched_rnms(b, s, l, 0.1),
               "v3 obb_batched": lambda: obb_batched_nms(b, s, l, 0.1),
               "v2 ml_nms": lambda: ml_nms_rotated(b, s, l, 0.1)}
        kept = batched_rnms(b, s, l, 0.1)[1].numel()
        for k, (med, mn) in time_variants(var, rounds=5, reps=5).items():
            print(f"nms  n={nn:6d} kept={kept:5d} {k:16s} med {med:9.1f} us  min {mn:9.1f} us  "
                  f"{nn / med:8.3f} Mboxes/s  {alg / med / 1e3:7.2f} GB/s", flush=True)


if __name__ == "__main__":
    _C.lib()
    what = sys.argv[1:] or ["iou", "fr", "nms"]
    if "iou" in what:
        bench_iou()
    if "assign" in what:
        bench_assign()
    if "fr" in what:
        bench_fr()
    if "nms" in what:
        bench_nms()
