#!/usr/bin/env python
"""bench.py -- R3Det R50-FPN 1024x1024 on MI355X with the MI355X-native rotated ops.

    python bench.py --gpus N --steps K --warmup W [--mode infer|train|rretinanet]      (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

--mode infer (default; BASELINE.json's metric).  One *step* = one full inference pass of
r3det_r50_fpn_1x (v1) over a batch of 4 synthetic 1024 x 1024 tiles per GPU (BASELINE configs[2];
configs[3] is the same step on 8 GPUs), inputs resident in HBM: ResNet-50 + FPN + RRetinaHead
(MIOpen convs, PyTorch-ROCm plumbing) -> filter_bboxes -> FeatureRefineModule (FR sampler =
libr3det_hip.so) -> RRetinaRefineHead -> per-image multiclass_nms_rotated, nms type 'v1'
(libr3det_hip.so).  Nothing is skipped or cached.  N > 1: image-parallel, every rank runs its own
batch, one RCCL all_gather of the padded detections per step (the only exchange of the path).
--mode train: BASELINE configs[4] -- one optimisation step of the same model (batch 2 x 1024^2 per
GPU, 128 synthetic GT per image): forward_train (fused MaxIoU assignment of 196 416 anchors and of
21 824 refined boxes per image, focal + smooth-L1 losses, FR sampler forward and packed backward
through autograd), backward, SGD; N > 1: DistributedDataParallel over RCCL.
--mode rretinanet: BASELINE configs[1] -- rretinanet_obb_r50_fpn v1 inference, batch 2 x 1024^2.

Rank 0 prints ONE JSON line of at most 1 800 characters (`headline`: metric, value, timing, config,
roofline, cpu_baseline) as the last line of stdout; everything else below is the detail record
bench_detail.json written next to this file (and under gpurun_out/).  stderr stays a few lines
(R3DET_BENCH_VERBOSE=1: the phase trace; R3DET_BENCH_DETAIL_STDERR=1: the detail record too).
`value` = images/s over all ranks.  `hot_path` repeats the
measurement for the custom ops alone (same shapes, no convs).  `roofline` is for the dominant
HBM-bound hand-written kernel (FR forward, level 0), timed with HIP events attached to the launch
inside the timed region.  `cpu_baseline` times the oracle / oracle/_ref on a bounded sample of the
hot path on the host cores (rank 0, N = 1 only), with per-op rows beside it.  `ops` carries the
op-level rates BASELINE.json names (rotated-IoU Mpairs/s, NMS Mboxes/s), each with its own roofline
entry.  In the default mode rank 0 of a single-GPU run also adds bounded `train` (configs[4]) and
`rretinanet` (configs[1]) entries.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402

BATCH = 4
TRAIN_BATCH = 2       # configs/_base_/datasets/dota1_0.py:31 samples_per_gpu=2
TRAIN_GT = 128
RRETINA_BATCH = 2     # BASELINE configs[1]
C = 256
IMG = 1024
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (~6.3 TB/s achievable)
NMS_CFG = dict(iou_thr=0.1)  # type absent -> 'v1' (bbox_nms_rotated.py:43)
SCORE_THR, MAX_PER_IMG = 0.05, 2000
CHANNELS_LAST = os.environ.get("R3DET_BENCH_NCHW", "0") != "1"  # activation layout of the conv stack
# R3DET_TRAIN_CHANNELS_LAST=1: the training step in channels_last -- the FR sampler and its backward then run on NHWC
# memory (r3det_feature_refine_backward_nhwc).  Off by default: MIOpen's fp32 backward convolutions are slower on
# NHWC activations (62.4 vs 53.5 ms per step measured, round 3), although the FR backward itself is faster there.
TRAIN_CHANNELS_LAST = os.environ.get("R3DET_TRAIN_CHANNELS_LAST", "0") == "1"
FUSE = os.environ.get("R3DET_BENCH_NOFUSE", "0") != "1"          # conv+BN folding and fused epilogues
PROFILE_PMC = os.path.join(ROOT, "profiles", "roofline_kernel_pmc.json")
FR_SOURCE = os.path.join(ROOT, "r3det-pytorch_amd", "csrc", "r3_fr.hip")


def sha16(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]


# ------------------------------------------------------------------------------------ inference
def build_model(device, seed, kind="R3Det", batch=BATCH, spread=True):
    from r3det.models import R3Det, RRetinaNet
    from r3det.models.detectors import calibrate_score_bias
    torch.manual_seed(seed)
    model = (R3Det if kind == "R3Det" else RRetinaNet)().eval().to(device)
    if FUSE:
        # what the reference's benchmark does with --fuse-conv-bn (tools/analysis_tools/benchmark.py:88-89),
        # plus one-pass bias / ReLU / residual epilogues (r3det_bias_act) instead of 2-3 elementwise launches
        from r3det.models.fuse import fuse_for_inference
        fuse_for_inference(model)
    g = torch.Generator(device="cpu")
    g.manual_seed(seed + 1)
    img = torch.randn(batch, 3, IMG, IMG, generator=g).to(device)
    if CHANNELS_LAST:
        # MIOpen's fp32 convolutions run ~10 % faster on NHWC activations (tools/cl_probe.py)
        model = model.to(memory_format=torch.channels_last)
        if os.environ.get("R3DET_BENCH_FRM_NCHW", "0") == "1":
            for m in getattr(model, "feat_refine_module", []):
                m.to(memory_format=torch.contiguous_format)
        img = img.contiguous(memory_format=torch.channels_last)
    # ~3.3 k NMS candidates / image; spread: every one of the 15 classes the same share of them (the
    # DOTA-shaped pool); otherwise ONE common shift, which leaves ~90 % of a random head's candidates in one class
    calibrate_score_bias(model, img, frac=0.01, per_class=spread)
    return model, img


def model_step(model, img, batch_size=None):
    """One inference step of a rank and the step's ONE exchange.  ``batch_size``: the configured per-rank batch (every
    rank passes the same value; default: this rank's batch -- the bench's ranks all run the same synthetic batch): a
    rank's short last batch is padded to it inside gather_detections with no extra collective and no host read."""
    from r3det import dist_infer as di
    res = model.simple_test(img)
    packed, counts = di.pack_detections([r[0] for r in res], [r[1] for r in res], MAX_PER_IMG)
    di.gather_detections(packed, counts, batch_size=img.size(0) if batch_size is None else batch_size)
    return counts


class WholeStep:
    """The timed inference step: R3Det.simple_test as ONE HIP graph (network, decoding, pool, multiclass NMS, padded
    result: models/detectors.py::GraphedStep) + the step's one exchange on the buffer the graph wrote
    (dist_infer.gather_padded).  The capacity's overflow flags are looked at two steps late (GraphedStep, lag 2): the
    host only ever waits for a step the GPU finished a whole step ago, so it stays one step ahead of the device; a step
    that overflowed is run again, inside the timed region, after the graph was recorded anew."""

    def __init__(self, model, img):
        from r3det.models.detectors import GraphedStep
        self.g = GraphedStep(model, img)
        self.img = img
        self.redone = 0

    def __call__(self):
        from r3det import dist_infer as di
        out, redo = self.g.step(self.img)
        if redo:  # one of the last `redo` steps outgrew the capacity (the graph now has twice the room): those steps again,
            #       inside the timed region -- the image is the same every step, so "again" is `redo` more steps
            self.redone += redo
            for _ in range(redo):
                out, _ = self.g.step(self.img)
        di.gather_padded(out)
        return out

    def counts(self):
        if self.g.flush():               # (the last steps of the loop were not looked at yet)
            self.redone += 1
            self.g.step(self.img)
        return [int(c) for c in self.g.static_out[:, -1, 0].tolist()]


def host_results(out, num_classes=15):
    """What the reference's timed call ends in (tools/analysis_tools/benchmark.py:104-110 -> simple_test ->
    rbbox2result, models/detectors/r3det.py:137-143, core/bbox/rtransforms.py:10-25): detections to the host, one
    ndarray per class and image."""
    from r3det.core.bbox.rtransforms import rbbox2result
    o = out.cpu()
    rows = o.size(1) - 1
    res = []
    for i in range(o.size(0)):
        k = int(o[i, rows, 0])
        res.append(rbbox2result(o[i, :k, :6], o[i, :k, 6].long(), num_classes))
    return res


def device_time_ms(fn, reps=10, lead_ms=None):
    """GPU time of ``fn`` per call with the HOST out of the picture: a spin kernel keeps the stream busy while the
    host enqueues `reps` calls, so the device runs them back to back; events around them.  (The wall figure of a
    launch-wait-launch loop next to it says how much of an operator's time is the host's.)  The spin lasts 1.5 x what
    the host needed for the same `reps` calls a moment before, so the queue cannot run dry behind it."""
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    host_ms = (time.perf_counter() - t0) * 1e3  # (enqueue time only: no synchronize inside)
    torch.cuda.synchronize()
    if lead_ms is None:
        lead_ms = max(2.0, 1.5 * host_ms)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(int(lead_ms * 2.0e6))  # (~2 GHz cycles)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def spread_stats(samples_s):
    """median / min / max of wall samples (seconds) in ms."""
    srt = sorted(samples_s)
    return {"median": round(srt[len(srt) // 2] * 1e3, 3), "min": round(srt[0] * 1e3, 3), "max": round(srt[-1] * 1e3, 3),
            "n": len(srt)}


# ------------------------------------------------------------------------------------ training
def build_train(device, seed, world, channels_last=None):
    # MIOpen's exhaustive find mode (cudnn.benchmark) searches every convolution of the step in three directions:
    # 9 minutes on a fresh box for a 57 ms step.  The training step runs in immediate mode.
    torch.backends.cudnn.benchmark = False
    from r3det import dist_train as dt
    from r3det import synthetic as syn
    from r3det.models import R3Det
    torch.manual_seed(seed)
    channels_last = TRAIN_CHANNELS_LAST if channels_last is None else channels_last
    model = R3Det().train().to(device)
    if channels_last:
        model = model.to(memory_format=torch.channels_last)
    ddp = dt.wrap_ddp(model, device) if world > 1 else model
    opt = dt.build_optimizer(model)
    g = torch.Generator(device="cpu")
    g.manual_seed(seed + 1)
    img = torch.randn(TRAIN_BATCH, 3, IMG, IMG, generator=g).to(device)
    if channels_last:
        img = img.contiguous(memory_format=torch.channels_last)
    gtb = [syn.dota_like_rboxes(TRAIN_GT, seed * 10 + i, device=device) for i in range(TRAIN_BATCH)]
    gtl = [torch.randint(0, 15, (TRAIN_GT,), generator=g).to(device) for _ in range(TRAIN_BATCH)]
    return dict(model=model, ddp=ddp, opt=opt, img=img, gtb=gtb, gtl=gtl, channels_last=channels_last)


def train_step(tr):
    from r3det import dist_train as dt
    return dt.train_step(tr["ddp"], tr["opt"], tr["img"], tr["gtb"], tr["gtl"])[0]


def train_custom_op_ms(tr, device):
    """The custom-op part of the step on its own, same shapes: fused assignment of both stages for both images,
    FR sampler forward + packed backward over the pyramid (no convolutions, no losses)."""
    from r3det import synthetic as syn
    from r3det.core import obb2hbb
    from r3det.ops.feature_refine import feature_refine_levels
    m = tr["model"]
    anchors = torch.cat(m.bbox_head.anchors([(IMG // s, IMG // s) for s in syn.STRIDES], device))
    feats, boxes = syn.fr_pyramid(TRAIN_BATCH, C, 9, device=device)
    refined = [torch.cat([b.view(TRAIN_BATCH, -1, 5)[i] for b in boxes]) for i in range(TRAIN_BATCH)]
    if tr["channels_last"]:
        feats = [f.contiguous(memory_format=torch.channels_last) for f in feats]
    xs = [f.clone().requires_grad_(True) for f in feats]  # (clone preserves the layout)
    gs = [torch.randn_like(f) for f in feats]

    def assign():
        for i in range(TRAIN_BATCH):
            # (the anchor head names its grid: the assigner keeps the grid's prepared columns, as in the step itself)
            m.bbox_head.assigner.assign(anchors, obb2hbb(tr["gtb"][i], 'v1'), None, tr["gtl"][i],
                                        shared_key=('bench_anchor_grid', anchors.data_ptr()))
            m.refine_head[0].assigner.assign(refined[i], tr["gtb"][i], None, tr["gtl"][i])

    from r3det.ops.feature_refine import feature_refine_module_levels
    as_ = [torch.randn_like(f).requires_grad_(True) for f in feats]  # (the module's two convolution outputs)
    bs_ = [torch.randn_like(f).requires_grad_(True) for f in feats]
    scales = [1.0 / s for s in syn.STRIDES]

    def fr():
        for t in xs + as_ + bs_:
            t.grad = None
        # what FeatureRefineModule runs in training (round 5): add, samplers and residual add of the five levels as ONE
        # autograd node (both layouts)
        torch.autograd.backward(feature_refine_module_levels(as_, bs_, xs, boxes, scales, 1), gs)

    def fr_three_step():
        for t in xs + as_ + bs_:
            t.grad = None
        # rounds 3-4: the samplers as one node, the two elementwise passes per level around it
        sampled = feature_refine_levels([a + b for a, b in zip(as_, bs_)], boxes, scales, 1)
        torch.autograd.backward([x + o for x, o in zip(xs, sampled)], gs)
    def samples(fn, n=9, reps=5):  # n wall figures of `reps` calls each
        fn()
        out = []
        for _ in range(n):
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            out.append((time.perf_counter() - t) / reps)
        return out
    class _NullNode(torch.autograd.Function):
        """A node of the SAME signature with no library call (outputs and gradients allocated, nothing launched): what
        torch's autograd machinery costs for fr() -- apply with 23 arguments, the engine's thread hand-off, 15 leaf
        accumulations -- i.e. the floor under fr_fwd_bwd_ms_wall that no host code of this repository can go below."""
        @staticmethod
        def forward(ctx, scales_, points_, n, *tensors):
            ctx.n = n
            return tuple(torch.empty_like(t) for t in tensors[2 * n:3 * n])

        @staticmethod
        def backward(ctx, *grads):
            ds = tuple(torch.empty_like(g) for g in grads)
            return (None, None, None) + ds + ds + tuple(grads) + (None,) * ctx.n

    def fr_null():
        for t in xs + as_ + bs_:
            t.grad = None
        torch.autograd.backward(_NullNode.apply(tuple(scales), 1, len(xs), *as_, *bs_, *xs, *boxes), gs)
    sa, sf, s3, s0 = samples(assign), samples(fr), samples(fr_three_step), samples(fr_null)
    detail = {"what": "wall = launch-wait-launch loops of 5 calls, median / min / max of 9; device = GPU time of the same "
                      "calls with the stream kept busy while the host enqueues them (HIP events): the difference is the "
                      "host's share",
              "assign_ms_wall": spread_stats(sa), "assign_ms_device": round(device_time_ms(assign, reps=5), 3),
              "fr_fwd_bwd_ms_wall": spread_stats(sf), "fr_fwd_bwd_ms_device": round(device_time_ms(fr, reps=5), 3),
              "fr_what": "module tail of the five levels (add, samplers, residual add; backward: the gathers) as one autograd "
                         "node; three_step = the same work as rounds 3-4 ran it (elementwise adds outside the node)",
              "fr_null_node_ms_wall": spread_stats(s0),
              "fr_null_node_what": "an autograd node of the same signature that launches nothing: torch's own cost for this "
                                   "call pattern (the floor under fr_fwd_bwd_ms_wall)",
              "fr_three_step_ms_wall": spread_stats(s3),
              "fr_three_step_ms_device": round(device_time_ms(fr_three_step, reps=5), 3)}
    train_custom_op_ms.detail = detail
    return sorted(sa)[len(sa) // 2] * 1e3, sorted(sf)[len(sf) // 2] * 1e3


# ------------------------------------------------------------------------------------ custom ops alone
def build_hot_workload(device, seed):
    """What the custom-op layer of R3Det.simple_test runs per step at BATCH x 1024^2 in the channels_last model:
    per pyramid level the FeatureRefineModule tail (one fr_module_nhwc launch), the refine head's pre-NMS pool of
    all levels (r3det_levels_pool: levels 0 / 1 cut at nms_pre = 2000 -> 5344 rows per image), the batched multiclass
    NMS (v1).  Synthetic head maps: ~4 % of the (row, class) scores pass score_thr, as in the calibrated model."""
    from r3det import synthetic as syn
    from r3det.core.post_processing import CapacityHint
    cl = torch.channels_last
    g = torch.Generator(device=device).manual_seed(seed)
    feats, boxes = syn.fr_pyramid(BATCH, C, seed, device=device)
    levels = []
    for f, bx, st in zip(feats, boxes, syn.STRIDES):
        H, W = f.shape[-2:]

        def mk(*shape, scale=1.0, shift=0.0):
            return (torch.randn(*shape, device=device, generator=g) * scale + shift).contiguous(memory_format=cl)
        levels.append(dict(a=mk(*f.shape), b=mk(*f.shape), res=f.contiguous(memory_format=cl),
                           out=torch.empty_like(f, memory_format=cl), boxes=bx, scale=1.0 / st,
                           cls=mk(BATCH, 15, H, W, scale=1.5, shift=-5.5), reg=mk(BATCH, 5, H, W, scale=0.1),
                           rois=bx.view(BATCH, H * W, 5).contiguous(), rows=min(2000, H * W)))
    n = sum(lv["rows"] for lv in levels)
    return dict(levels=levels, bias=torch.randn(C, device=device, generator=g),
                pool_boxes=torch.empty(BATCH, n, 5, device=device), pool_scores=torch.empty(BATCH, n, 16, device=device),
                nms_hint=CapacityHint(), pnms=None)


def hot_path_step(wl):
    from r3det.core.post_processing import multiclass_nms_rotated_batch
    from r3det.ops import fr_boxes
    from r3det.ops.feature_refine import fr_module_levels_nhwc
    L = wl["levels"]
    # (what FeatureRefineModule.forward runs: the module tail of all levels in one library call)
    fr_module_levels_nhwc([lv["a"] for lv in L], [lv["b"] for lv in L], wl["bias"], wl["bias"], [lv["res"] for lv in L],
                          [lv["boxes"] for lv in L], [lv["scale"] for lv in L], 1, [lv["out"] for lv in L])
    fr_boxes.levels_pool([lv["cls"] for lv in L], [lv["reg"] for lv in L], [lv["rois"] for lv in L], 1, 15, 2000,
                         (IMG, IMG), wl["pool_boxes"], wl["pool_scores"])
    res = multiclass_nms_rotated_batch(wl["pool_boxes"], wl["pool_scores"], SCORE_THR, NMS_CFG, MAX_PER_IMG,
                                       hint=wl["nms_hint"])
    return sum(d.size(0) for d, _ in res)


def hot_path_step_sync_free(wl):
    """The same three calls with the padded NMS result (r3det_mcnms_padded): no host read anywhere."""
    from r3det.core.post_processing import PaddedNms
    from r3det.ops import fr_boxes
    from r3det.ops.feature_refine import fr_module_levels_nhwc
    L = wl["levels"]
    fr_module_levels_nhwc([lv["a"] for lv in L], [lv["b"] for lv in L], wl["bias"], wl["bias"], [lv["res"] for lv in L],
                          [lv["boxes"] for lv in L], [lv["scale"] for lv in L], 1, [lv["out"] for lv in L])
    fr_boxes.levels_pool([lv["cls"] for lv in L], [lv["reg"] for lv in L], [lv["rois"] for lv in L], 1, 15, 2000,
                         (IMG, IMG), wl["pool_boxes"], wl["pool_scores"])
    if wl["pnms"] is None:  # (first call: the capacity from this pool, with room)
        m = int((wl["pool_scores"][..., :-1] > SCORE_THR).flatten(1).sum(1).max().item())
        wl["pnms"] = PaddedNms(BATCH, wl["pool_boxes"].size(1), 15, SCORE_THR, NMS_CFG, MAX_PER_IMG, int(m * 1.3),
                               wl["pool_boxes"].device)
    return wl["pnms"](wl["pool_boxes"], wl["pool_scores"])


def timeit(fn, reps, warm=2):
    for _ in range(warm):
        fn()
    best = float("inf")
    for _ in range(3):  # best of three batches
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t) / reps)
    return best


def b_iou(m, n):
    return 4 * m * n + 20 * (m + n)


def b_nms(n):
    w = (n + 63) // 64
    return 24 * n + 8 * n * w + 8 * n        # boxes + upper-triangle mask written and read + keep


def _roof(nbytes, dt, bound="hbm"):
    gbs = nbytes / dt / 1e9
    return {"bound": bound, "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(gbs / HBM_PEAK_GBS, 5)}


def b_fr(N, hw, points=1):
    return 2 * 4 * N * C * hw + 20 * N * hw     # one read + one write per element, one box per position


def op_rates(device):
    """Op-level rates quoted by BASELINE.json's metric (Mpairs/s, Mboxes/s) and SURVEY 8d's micro-benchmark list,
    one roofline entry per row: algorithmic bytes / time per call (wall clock around back-to-back
    calls, synchronised, so host launch gaps are inside; the per-kernel durations of the same calls are in
    profiles/r03_*).  FR rows rotate over buffer sets of >= 0.6 GB where the level is large enough, so their inputs
    come from HBM, not from the 256 MiB Infinity Cache."""
    from r3det import synthetic as syn
    from r3det.ops import batched_rnms, obb_batched_nms, obb_overlaps, rbbox_iou
    out = {}
    anchors = syn.anchor_grid(device=device)
    gt = syn.dota_like_rboxes(128, 5, device=device)
    gt512 = syn.dota_like_rboxes(512, 6, device=device)
    refined = torch.cat([syn.fr_level_boxes(1, IMG // s, IMG // s, s, 50 + i, device=device)
                         for i, s in enumerate(syn.STRIDES)])
    a, g = syn.rand_rboxes(1000, 0, device=device), syn.rand_rboxes(128, 1, device=device)
    for name, fn, b1, b2, reps in (("iou_v1_128x196416", rbbox_iou, gt, anchors, 20),
                                   ("iou_v1_512x196416", rbbox_iou, gt512, anchors, 10),
                                   ("iou_v1_128x21824", rbbox_iou, gt, refined, 50),
                                   ("iou_v1_1000x128", rbbox_iou, a, g, 50),
                                   ("iou_v3_128x196416", obb_overlaps, gt, anchors, 10),
                                   ("iou_v3_1000x128", obb_overlaps, a, g, 50)):
        dt = timeit(lambda: fn(b1, b2), reps)
        m, n = b1.size(0), b2.size(0)
        dev_us = device_time_ms(lambda: fn(b1, b2), reps=10) * 1e3  # GPU time, the host out of the picture
        out[name] = {"Mpairs_s": round(m * n / dt / 1e6, 1), "us_per_call": round(dt * 1e6, 2),
                     "us_per_call_device": round(dev_us, 2),
                     "alg_bytes": b_iou(m, n), "roofline": _roof(b_iou(m, n), dt),
                     "roofline_device": _roof(b_iou(m, n), dev_us * 1e-6)}
    for n, fn, tag in ((2000, batched_rnms, "v1"), (5344, batched_rnms, "v1"), (8576, batched_rnms, "v1"),
                       (32768, batched_rnms, "v1"), (8576, obb_batched_nms, "v3")):
        mb, ms = syn.nms_pool(n * 10 // 6 + 64, 77 + n, device=device)
        sc, lab = ms[:, :-1].max(1)
        idx = torch.nonzero(sc > SCORE_THR).squeeze(1)[:n]
        b, s, l = mb[idx].contiguous(), sc[idx].contiguous(), lab[idx].contiguous()
        dt = timeit(lambda: fn(b, s, l, 0.1), 10)
        k = b.size(0)
        out[f"nms_{tag}_{k}"] = {"Mboxes_s": round(k / dt / 1e6, 3), "us_per_call": round(dt * 1e6, 1),
                                 "alg_bytes": b_nms(k),
                                 "what": f"{fn.__name__} (15 classes) incl. its host read of the count",
                                 "roofline": _roof(b_nms(k), dt, "hbm (latency-bound in practice)")}
        # the same library call without the host read of the count (r3det.ops.nms.batched_rnms_padded): back-to-back
        # calls keep the queue full, so this is the call's GPU time + launch gaps -- what it costs inside a pipeline
        from r3det.ops.nms import batched_rnms_padded
        if batched_rnms_padded(b, s, l, 0.1, version=tag) is not None:
            dtp = timeit(lambda: batched_rnms_padded(b, s, l, 0.1, version=tag), 10)
            out[f"nms_{tag}_{k}"]["us_per_call_padded"] = round(dtp * 1e6, 1)
    out.update(pool_rates(device))
    out.update(fr_rates(device))
    return out


def pool_rates(device):
    """The pre-NMS pool of a whole head in one library call (r3det_levels_pool: sigmoid, per-image top-2000 per level,
    decode) at the two BASELINE models' shapes; algorithmic bytes = the head's cls + reg maps read once + the pool
    arrays written once."""
    from r3det.ops import fr_boxes
    g = torch.Generator(device=device).manual_seed(3)
    out, Cc, k = {}, 15, 2000
    for name, N, A in (("pool_r3det_refine_N4", 4, 1), ("pool_rretinanet_N2", 2, 9)):
        sizes = [IMG // s for s in (8, 16, 32, 64, 128)]
        cl = torch.channels_last
        cls = [(torch.randn(N, A * Cc, H, H, device=device, generator=g) * 1.5 - 4.0).contiguous(memory_format=cl) for H in sizes]
        reg = [(torch.randn(N, A * 5, H, H, device=device, generator=g) * 0.2).contiguous(memory_format=cl) for H in sizes]
        anc = [torch.rand(H * H * A, 5, device=device, generator=g) * 50 + 5 for H in sizes]
        n = sum(min(k, H * H * A) for H in sizes)
        boxes, scores = torch.empty(N, n, 5, device=device), torch.empty(N, n, Cc + 1, device=device)
        dt = timeit(lambda: fr_boxes.levels_pool(cls, reg, anc, A, Cc, k, (IMG, IMG), boxes, scores), 20)
        nb = sum(c.numel() + r.numel() for c, r in zip(cls, reg)) * 4 + (boxes.numel() + scores.numel()) * 4
        out[name] = {"us_per_call": round(dt * 1e6, 1), "alg_bytes": nb, "pool_rows_per_image": n,
                     "what": f"r3det_levels_pool, five levels, A = {A}, nms_pre = {k}: memset + 3 launches",
                     "roofline": _roof(nb, dt, "hbm (launch- and latency-bound in practice)")}
    return out


def fr_rates(device):
    """FR forward per level and fused-5-level, N in {1, 2, 4, 8}, points in {1, 5}; FR backward with the
    same B_fr.  NCHW = the reference's layout (r3det_feature_refine_forward / _backward_ws), NHWC = channels_last
    (r3det_feature_refine_forward_nhwc / _backward_nhwc)."""
    from r3det import synthetic as syn
    from r3det.ops import feature_refine as FRM
    cl = torch.channels_last
    out = {}

    def run(name, N, lvls, points, fn, nhwc, reps=10, prep=None, what=None):
        feats, boxes = syn.fr_pyramid(N, C, 31, device=device)
        feats, boxes = [feats[i] for i in lvls], [boxes[i] for i in lvls]
        scales = [1.0 / syn.STRIDES[i] for i in lvls]
        extra = prep(feats, boxes, scales) if prep else None
        per_set = 2 * 4 * sum(f.numel() for f in feats)
        nset = max(2, min(16, int(6e8 // per_set) + 1))
        sets = []
        for _ in range(nset):
            xs = [torch.randn_like(f) for f in feats]
            if nhwc:
                xs = [x.contiguous(memory_format=cl) for x in xs]
            sets.append((xs, [torch.empty_like(x) for x in xs]))
        state = [0]

        def call():
            xs, os_ = sets[state[0] % nset]
            state[0] += 1
            if prep:
                fn(xs, boxes, scales, points, os_, extra)
            else:
                fn(xs, boxes, scales, points, os_)
        dt = timeit(call, reps)
        hw = sum(f.shape[-1] * f.shape[-2] for f in feats)
        nb = b_fr(N, hw, points)
        out[name] = {"us_per_call": round(dt * 1e6, 1), "alg_bytes": nb, "library_calls": 1 if len(lvls) > 1 else len(lvls),
                     "rotating_MB": round(per_set * nset / 1e6), "roofline": _roof(nb, dt)}
        if what:
            out[name]["what"] = what
        del sets

    # one level: the per-level entry points; the five levels: ONE library call (what FeatureRefineModule runs)
    def fwd_nchw(xs, bs, scs, p, os_):
        if len(xs) == 1:
            FRM.fr_forward(xs[0], bs[0], scs[0], p, os_[0])
        else:
            FRM.fr_forward_levels(xs, bs, scs, p, os_)

    def fwd_nhwc(xs, bs, scs, p, os_):
        if len(xs) == 1:
            FRM.fr_forward_nhwc(xs[0], bs[0], scs[0], p, os_[0])
        else:
            FRM.fr_forward_levels_nhwc(xs, bs, scs, p, os_)

    def bwd_nchw(xs, bs, scs, p, os_):  # (index + gather: the whole backward op, as the reference's one call is)
        if len(xs) == 1:
            FRM.fr_backward(xs[0], bs[0], scs[0], p, os_[0], overwrite=True)
        else:
            FRM.fr_backward_levels(xs, bs, scs, p, os_)

    def bwd_nhwc(xs, bs, scs, p, os_):
        if len(xs) == 1:
            FRM.fr_backward_nhwc(xs[0], bs[0], scs[0], p, os_[0], overwrite=True)
        else:
            FRM.fr_backward_levels_nhwc(xs, bs, scs, p, os_)

    # round 6: the backward as a TRAINING step runs it on channels_last memory -- the forward launches of the same boxes
    # left the levels' tap tables behind (r3det_feature_refine_*_levels_nhwc_tab; built once here, outside the timed
    # calls, as the forward pass is), the index kernel scans those, then the gathers
    def tables_of(feats, boxes, scales):
        fcl = [f.contiguous(memory_format=cl) for f in feats]
        shapes = [tuple(f.shape[2:]) for f in fcl]
        tabs = FRM.tap_tables(fcl[0].size(0), shapes, device)
        assert FRM.fr_forward_levels_nhwc(fcl, boxes, scales, 1, [torch.empty_like(f) for f in fcl], tabs)
        return tabs

    def bwd_nhwc_train(xs, bs, scs, p, os_, tabs):
        from r3det import _C as C_
        N_ = xs[0].size(0)
        shapes = [tuple(x.shape[2:]) for x in xs]
        ws, wsb = FRM.fr_backward_nhwc_index_levels(bs, N_, shapes, scs, 1, tabs)
        P = FRM._plan(N_, 0, shapes, scs, 1)
        C_.check(C_.lib().r3det_feature_refine_backward_nhwc_levels_indexed(
            len(xs), FRM._ptr_array(xs), N_, xs[0].size(1), P.H, P.W, 1, FRM._ptr_array(os_), 1, C_.ptr(ws), wsb,
            C_.stream()), "fr_backward_nhwc_levels_indexed")

    fwd, bwd = {False: fwd_nchw, True: fwd_nhwc}, {False: bwd_nchw, True: bwd_nhwc}
    for nhwc, lay in ((False, "nchw"), (True, "nhwc")):
        for points in (1, 5):
            for lvl in range(5):
                run(f"fr_fwd_{lay}_p{points}_N4_L{lvl}", 4, [lvl], points, fwd[nhwc], nhwc)
            for N in (1, 2, 4, 8):
                run(f"fr_fwd_{lay}_p{points}_N{N}_5lvl", N, list(range(5)), points, fwd[nhwc], nhwc)
        for N in (2, 4):
            for lvls, tag in (([0], "L0"), ([1], "L1"), (list(range(5)), "5lvl")):
                run(f"fr_bwd_{lay}_p1_N{N}_{tag}", N, lvls, 1, bwd[nhwc], nhwc)
    for N in (2, 4):
        for lvls, tag in (([0], "L0"), (list(range(5)), "5lvl")):
            run(f"fr_bwd_nhwc_p1_N{N}_{tag}_train", N, lvls, 1, bwd_nhwc_train, True, prep=tables_of,
                what="the backward as a training step runs it: index from the tap tables the forward launches wrote "
                     "(2 library calls: index, gathers)")
    return out


# ------------------------------------------------------------------------------------ host baselines
def cpu_baseline():
    """Hot path of ONE image on the host: FR forward (oracle, OpenMP) + NMS v1 (reference CPU code from
    oracle/_ref when present, else the oracle); `per_op`: the op-level rows of BASELINE.md section 3."""
    import numpy as np
    from oracle import api as O
    from r3det import synthetic as syn
    cores = max(1, len(os.sched_getaffinity(0)))
    feats, boxes = syn.fr_pyramid(1, C, 1234)
    mb, ms = syn.nms_pool(syn.R3DET_POOL, 4321)
    mbn, msn = mb.numpy(), ms.numpy()
    t0 = time.perf_counter()
    reps = 0
    use_ref = O.ref_available()
    while True:
        for f, b, s in zip(feats, boxes, syn.STRIDES):
            O.fr_forward(f.numpy(), b.numpy(), 1.0 / s, 1, threads=cores)
        sc = msn[:, :-1]
        valid = sc > SCORE_THR
        idx = np.argwhere(valid)
        bx, scv, lab = mbn[idx[:, 0]], sc[valid], idx[:, 1]
        sh = bx.copy()
        sh[:, :2] += (lab.astype(np.float32) * (bx.max() + 1))[:, None]
        if use_ref:
            O.ref_v1_rnms(np.hstack([sh, scv[:, None]]), 0.1)
        else:
            O.nms(O.V1, sh, scv, 0.1, ascending=True)
        reps += 1
        if time.perf_counter() - t0 > 8 or reps >= 20:
            break
    dt = (time.perf_counter() - t0) / reps

    def clock(fn, budget=1.5):
        fn()
        t, n = time.perf_counter(), 0
        while True:
            fn()
            n += 1
            if time.perf_counter() - t > budget or n >= 5:
                return (time.perf_counter() - t) / n
    per_op = {}
    a, g = syn.rand_rboxes(1000, 0).numpy(), syn.rand_rboxes(128, 1).numpy()
    anchors, gt = syn.anchor_grid().numpy(), syn.dota_like_rboxes(128, 5).numpy()
    sample = np.ascontiguousarray(anchors[::12])  # every 12th anchor of the grid: 16 368 columns
    if use_ref:
        t = clock(lambda: O.ref_v1_iou_mat(a, g))
        per_op["iou_v1_1000x128"] = {"Mpairs_s_1thread": round(128000 / t / 1e6, 3), "kind": "reference"}
        t = clock(lambda: O.ref_v1_iou_mat(gt, sample))
        per_op["iou_v1_128x196416"] = {"Mpairs_s_1thread": round(128 * len(sample) / t / 1e6, 3), "kind": "reference",
                                       "sample": f"128 x {len(sample)} (every 12th anchor)"}
    # OpenMP over rows: no more threads than rows, and at most 64 for the 128 000-pair case (256 threads on it
    # measured thread start-up, not the op: 1.03 Mpairs/s against 11.4 on one thread)
    th = min(cores, 64)
    t = clock(lambda: O.iou_mat(O.V1, a, g, threads=th))
    per_op.setdefault("iou_v1_1000x128", {})[f"Mpairs_s_openmp_{th}"] = round(128000 / t / 1e6, 3)
    th = min(cores, 128)
    t = clock(lambda: O.iou_mat(O.V1, gt, anchors, threads=th))
    per_op.setdefault("iou_v1_128x196416", {})[f"Mpairs_s_openmp_{th}"] = round(128 * len(anchors) / t / 1e6, 3)
    for n in (2000, 5344, 8576):
        pb, ps = syn.nms_pool(n * 10 // 6 + 64, 77 + n)
        sc, lab = ps[:, :-1].max(1)
        idx = torch.nonzero(sc > SCORE_THR).squeeze(1)[:n]
        b, s, l = pb[idx].numpy(), sc[idx].numpy(), lab[idx].numpy()
        sh = b.copy()
        sh[:, :2] += (l.astype(np.float32) * (b.max() + 1))[:, None]
        d6 = np.hstack([sh, s[:, None]])
        t = clock(lambda: O.ref_v1_rnms(d6, 0.1) if use_ref else O.nms(O.V1, sh, s, 0.1, ascending=True))
        per_op[f"nms_v1_{len(b)}"] = {"Mboxes_s_1thread": round(len(b) / t / 1e6, 4),
                                      "kind": "reference" if use_ref else "port"}
    # BASELINE.md section 3's v3 / v2 rows: the reference's own CPU code (oracle/_ref), one thread, bounded -- v3 IoU
    # 1000 x 128 (~0.5 s per call), v3 and v2 (15 labels) NMS at 2000 / 8576: one timed call each
    if use_ref:
        def once(fn):
            t = time.perf_counter()
            fn()
            return time.perf_counter() - t
        t = once(lambda: O.ref_v3_iou_mat(a, g))
        per_op["iou_v3_1000x128"] = {"Mpairs_s_1thread": round(128000 / t / 1e6, 3), "kind": "reference", "calls": 1}
        for n in (2000, 8576):
            pb, ps = syn.nms_pool(n * 10 // 6 + 64, 77 + n)
            sc, lab = ps[:, :-1].max(1)
            idx = torch.nonzero(sc > SCORE_THR).squeeze(1)[:n]
            b, s_, l = pb[idx].numpy(), sc[idx].numpy(), lab[idx].numpy()
            sh = b.copy()
            sh[:, :2] += (l.astype(np.float32) * (b.max() + 1))[:, None]
            t = once(lambda: O.ref_v3_nms(sh, s_, 0.1))
            per_op[f"nms_v3_{len(b)}"] = {"Mboxes_s_1thread": round(len(b) / t / 1e6, 4), "kind": "reference", "calls": 1}
            d6 = np.hstack([b, l.astype(np.float32)[:, None]])
            t = once(lambda: O.ref_v2_nms(d6, s_, 0.1))
            per_op[f"nms_v2_{len(b)}"] = {"Mboxes_s_1thread": round(len(b) / t / 1e6, 4), "kind": "reference", "calls": 1,
                                          "note": "the reference's label-aware IoU (ml_nms_rotated's header) under the "
                                                  "harness's greedy loop (oracle/ref_harness/harness_v2.cpp: the "
                                                  "reference's own loop goes through at::Tensor indexing and took 2.3 s "
                                                  "/ 22 s at 2000 / 8576 in the survey, BASELINE.md section 3)"}
    cpu_model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                cpu_model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": round(1.0 / dt, 3), "unit": "img/s", "cores": cores, "cpu_model": cpu_model,
            "kind": "port",
            "compare_with": "hot_path.img_s (the custom ops alone on the GPU; its launches also carry the module's "
                            "elementwise adds and the per-level top-k pool, which this CPU sample does not run), not "
                            "with `value` (full model)",
            "sample": f"{reps} x custom-op hot path of ONE image (no convs): FR forward 5 levels N=1 C=256 on "
                      f"{cores} threads [oracle port; the reference has no CPU FR] + NMS v1 on a 5344-box "
                      f"pool, 1 thread [{'reference rnms_cpu via oracle/_ref' if use_ref else 'oracle port'}]",
            "per_op": per_op}


def load_traffic():
    """HBM bytes per launch of the roofline kernel from the committed PMC passes -- only while the kernel
    source it was measured on is still the source of this build (the file records sha256(r3_fr.hip))."""
    if not os.path.exists(PROFILE_PMC):
        return None, "no PMC file"
    try:
        rec = json.load(open(PROFILE_PMC))
    except Exception:  # noqa: BLE001
        return None, "unreadable PMC file"
    if rec.get("kernel_source_sha16") != sha16(FR_SOURCE):
        return None, (f"stale: measured on r3_fr.hip {rec.get('kernel_source_sha16')} ({rec.get('commit')}), "
                      f"this build is {sha16(FR_SOURCE)}")
    return rec.get("hbm_bytes_per_launch"), f"{rec.get('kernel_symbol')} @ {rec.get('commit')}"


def _sync(device):
    if device.type == "cuda":  # (the gloo CPU tests drive this function with a CPU device)
        torch.cuda.synchronize()


def timed_region(step, args, device, di):
    """W untimed steps, then exactly K steps between barrier + synchronize; max over ranks."""
    for _ in range(args.warmup):
        step()
    _sync(device)
    di.barrier(device)
    _sync(device)
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = step()
    _sync(device)
    di.barrier(device)
    _sync(device)
    mine = time.perf_counter() - t0
    return di.max_over_ranks(mine, device), mine, last


def per_rank_ms(mine, steps, device, world):
    import torch.distributed as dist
    if world == 1 or not dist.is_initialized():
        return [round(mine / steps * 1e3, 3)]
    t = torch.tensor([mine / steps * 1e3], dtype=torch.float64, device=device)
    allt = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(allt, t)
    return [round(float(x), 3) for x in allt]


def dist_info():
    import torch.distributed as dist
    if dist.is_initialized():
        return {"backend": dist.get_backend(), "rccl_ranks": dist.get_world_size()}
    return {"backend": None, "rccl_ranks": 1}


_T0 = time.perf_counter()


_PHASES = []


def phase(name):
    """Wall-clock trace of the run (the driver clocks the whole command; this says where it went): kept for the detail
    record; on stderr only with R3DET_BENCH_VERBOSE=1 -- the driver's 2 000-character tail is stdout + stderr, and the
    headline has to stay inside it."""
    if int(os.environ.get("RANK", "0")) == 0:
        _PHASES.append([round(time.perf_counter() - _T0, 1), name])
        if os.environ.get("R3DET_BENCH_VERBOSE", "0") == "1":
            sys.stderr.write(f"[bench {time.perf_counter() - _T0:7.1f} s] {name}\n")
            sys.stderr.flush()


_ERRORS = {}


class section:
    """A part of the run that is NOT the timed region (op rates, the bounded train / rretinanet entries, the CPU baseline):
    traced like `phase`, and an exception inside is recorded in the detail record (`errors`) instead of costing the run its
    contract line -- the headline then carries what was measured before it."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        phase(self.name)
        return self

    def __exit__(self, et, ev, tb):
        if et is not None and issubclass(et, Exception):
            _ERRORS[self.name] = repr(ev)[:300]
            sys.stderr.write(f"[bench] section '{self.name}' failed: {repr(ev)[:200]}\n")
            return True
        return False


# ------------------------------------------------------------------------------------ the contract line
HEADLINE_MAX = 1800   # the driver keeps a 2 000-character tail of stdout + stderr; round 5's 22.8 KB line was not parsed
DETAIL_FILE = "bench_detail.json"
_HEAD_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data")
_ROOF_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "avg_launch_us", "traffic", "alg_bytes_per_launch")
_CPU_KEYS = ("value", "unit", "cores", "cpu_model", "kind", "sample")
_CFG_KEYS = ("workload", "batch_per_gpu", "global_batch", "nms_type")


def _short(s, n):
    s = str(s)
    return s if len(s) <= n else s[:n - 3] + "..."


def headline(line, detail_file=DETAIL_FILE):
    """The ONE stdout line of the contract, cut down to what the driver reads (<= HEADLINE_MAX characters): metric,
    value, timing, config, roofline, cpu_baseline.  Everything else the run measured (ops, hot_path, by_pool, train,
    rretinanet, per-op CPU rows, the notes) is the detail record `write_detail` puts next to bench.py.  The protocol
    the figure follows is the reference's tools/analysis_tools/benchmark.py:99-130 (one img/s figure)."""
    out = {k: line[k] for k in _HEAD_KEYS if k in line}
    cfg = line.get("config") or {}
    out["config"] = {k: cfg[k] for k in _CFG_KEYS if k in cfg}
    out["config"]["workload"] = _short(cfg.get("workload_short") or cfg.get("workload", ""), 160)
    roof = line.get("roofline")
    if roof is not None:
        r = {k: roof.get(k) for k in _ROOF_KEYS}
        r["kernel"] = _short(str(roof.get("kernel", "")).split(" = ")[0].split(" ")[0], 60)   # the symbol only
        out["roofline"] = r
    cpu = line.get("cpu_baseline")
    if cpu is not None:
        c = {k: cpu.get(k) for k in _CPU_KEYS}
        c["cpu_model"] = _short(c.get("cpu_model") or "", 48)
        c["sample"] = _short(c.get("sample") or "", 150)
        out["cpu_baseline"] = c
    for k in ("rccl_ranks", "per_rank_ms_per_step"):
        if k in line:
            out[k] = line[k]
    if isinstance(out.get("per_rank_ms_per_step"), list) and len(out["per_rank_ms_per_step"]) > 8:
        out["per_rank_ms_per_step"] = out["per_rank_ms_per_step"][:8]
    out["detail_file"] = detail_file
    # (never lose the line to its own size: shed the free text first, then the per-rank list -- the numbers stay)
    for shed in (None, ("cpu_baseline", "sample"), ("cpu_baseline", "cpu_model"), ("config", "workload"),
                 ("per_rank_ms_per_step",), ("metric",)):
        if shed is not None:
            if len(shed) == 2 and isinstance(out.get(shed[0]), dict) and shed[1] in out[shed[0]]:
                out[shed[0]][shed[1]] = _short(out[shed[0]][shed[1]], 40)
            elif len(shed) == 1 and shed[0] == "per_rank_ms_per_step" and isinstance(out.get(shed[0]), list):
                out[shed[0]] = out[shed[0]][:2]
            elif len(shed) == 1 and shed[0] in out:
                out[shed[0]] = _short(out[shed[0]], 80)
        text = json.dumps(out, separators=(", ", ": ")).replace("\n", " ")
        if len(text) <= HEADLINE_MAX:
            return text
    return text[:HEADLINE_MAX]  # (unreachable with the fields above; a cut line is still better than none)


def write_detail(line, name=DETAIL_FILE):
    """The full record (what rounds 1-5 printed as one line) as a file next to bench.py and, where the directory
    exists or can be made, under gpurun_out/ (the directory gpurun carries back).  Returns the paths written."""
    text = json.dumps(line, indent=1)
    done = []
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        try:
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, name), "w") as f:
                f.write(text + "\n")
            done.append(os.path.join(d, name))
        except OSError:
            pass
    return done


def emit(line):
    """Detail to the file, its location (one short line) to stderr, the compact headline as the LAST stdout line."""
    line = dict(line, phases_s=list(_PHASES))
    if _ERRORS:
        line["errors"] = dict(_ERRORS)
    paths = write_detail(line)
    text = headline(line)
    sys.stderr.write(f"[bench] detail ({len(json.dumps(line))} B): {', '.join(os.path.relpath(p, ROOT) for p in paths)}\n")
    if os.environ.get("R3DET_BENCH_DETAIL_STDERR", "0") == "1":
        sys.stderr.write(json.dumps(line) + "\n")
    sys.stderr.flush()
    sys.stdout.write(text + "\n")
    sys.stdout.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", choices=["infer", "train", "rretinanet"], default="infer")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ops", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the bounded train / rretinanet entries")
    ap.add_argument("--model-only", action="store_true",
                    help="stop after the timed model steps (used under rocprofv3: the tail of the trace is "
                         "then exactly the timed region)")
    args = ap.parse_args()
    if args.model_only:
        args.no_ops = args.no_cpu_baseline = args.no_extras = True

    from r3det import _C
    from r3det import dist_infer as di
    _C.lib()  # fail loudly if the HIP library is missing
    if os.environ.get("R3DET_NMS_IMPL"):  # (tools/nms_reducer_ab.sh: the reducer form of the batched NMS)
        _C.set_option("nms_impl", int(os.environ["R3DET_NMS_IMPL"]))
    rank, local_rank, world = di.env_world()
    if world == 1 and args.gpus > 1:
        raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    di.init(device=device)
    torch.backends.cudnn.benchmark = True  # MIOpen find mode for the backbone convs

    common = {"n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True,
              "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic"}

    if args.mode == "train":
        phase("train: build")
        tr = build_train(device, 300 + rank, world)
        elapsed, mine, loss = timed_region(lambda: train_step(tr), args, device, di)
        ranks = per_rank_ms(mine, args.steps, device, world)
        if rank == 0:
            ms = elapsed / args.steps * 1e3
            ta, tf = train_custom_op_ms(tr, device)
            line = dict(common, metric="img/s, R3Det R50-FPN 1024x1024 training step (r3det_r50_fpn_1x v1)",
                        value=round(world * TRAIN_BATCH * args.steps / elapsed, 2), unit="img/s",
                        ms_per_step=round(ms, 3), per_rank_ms_per_step=ranks, **dist_info(),
                        config={"workload_short": "BASELINE configs[4]: r3det_r50_fpn_1x v1 training step, batch 2 x 1024x1024 "
                                                  "per GPU, 128 GT per image, synthetic",
                                "workload": "BASELINE configs[4]: r3det_r50_fpn_1x v1 training step, batch=2 x 1024x1024 per "
                                            "GPU, 128 synthetic GT per image: forward_train (fused MaxIoU assignment, focal + "
                                            "smooth-L1, FR forward + backward), backward, SGD(momentum); random-init "
                                            "weights; norm_eval, frozen stem + layer1 as in the config",
                                "layout": "channels_last (FR sampler + backward on NHWC memory)" if TRAIN_CHANNELS_LAST
                                else "NCHW (packed FR backward)",
                                "batch_per_gpu": TRAIN_BATCH, "global_batch": TRAIN_BATCH * world,
                                "parallelism": f"DDP x{world} (RCCL all-reduce of 168 MB fp32 gradients in 48 MB buckets)"},
                        final_loss=round(float(loss), 4),
                        custom_ops={"what": "the step's custom ops timed on their own (isolated, same shapes)",
                                    "assign_ms": round(ta, 3), "fr_fwd_bwd_ms": round(tf, 3),
                                    "share_of_step": round((ta + tf) / ms, 4),
                                    "detail": getattr(train_custom_op_ms, "detail", None)})
            emit(line)
        if world > 1:
            di.barrier(device)
            torch.distributed.destroy_process_group()
        return

    if args.mode == "rretinanet":
        model, img = build_model(device, 200 + rank, "RRetinaNet", RRETINA_BATCH)
        whole = WholeStep(model, img)  # (round 5: the whole step as one HIP graph + gather_padded, as the main mode)
        elapsed, mine, _ = timed_region(whole, args, device, di)
        counts = torch.tensor(whole.counts())
        ranks = per_rank_ms(mine, args.steps, device, world)
        if rank == 0:
            emit(dict(
                common, metric="img/s, rretinanet_obb_r50_fpn v1 1024x1024 inference",
                value=round(world * RRETINA_BATCH * args.steps / elapsed, 2), unit="img/s",
                ms_per_step=round(elapsed / args.steps * 1e3, 3), per_rank_ms_per_step=ranks, **dist_info(),
                config={"workload_short": "BASELINE configs[1]: rretinanet_obb_r50_fpn v1, batch 2 x 1024x1024 per GPU, "
                                          "inference, synthetic",
                        "workload": "BASELINE configs[1]: rretinanet_obb_r50_fpn v1, batch=2 x 1024x1024 per GPU, inference "
                                    "(9 anchors / position, nms_pre 2000 per level -> 8576-box pools, nms v1)",
                        "batch_per_gpu": RRETINA_BATCH, "global_batch": RRETINA_BATCH * world},
                kept_per_image=[int(c) for c in counts.tolist()]))
        if world > 1:
            di.barrier(device)
            torch.distributed.destroy_process_group()
        return

    H = W = 128
    # The roofline launch is the FeatureRefineModule tail on channels_last memory (r3det_feature_refine_module_nhwc):
    # conv_a, conv_b and the residual read once, the output written once = 16 B per element (SURVEY 8d's 8 B per
    # element for the bare sampler + the module's two extra input streams that the launch folds in), plus the
    # 20-byte box per position.
    alg_bytes = 4 * 4 * BATCH * C * H * W + 20 * BATCH * H * W

    # ---- the custom ops on their own, BEFORE the model exists (round 1 / 2 measured them after `del model;
    # empty_cache()` and twice met one ~85 ms host-side launch stall right there: tools/hip_trace_slow.py)
    hot, alone_rec, ops = None, None, None
    if rank == 0 and not args.model_only:
        with section("hot path (custom ops alone)"):
            wl = build_hot_workload(device, seed=7)
            per, allocs = [], []
            for i in range(3 + 30):
                s0 = torch.cuda.memory_stats(device)["num_device_alloc"]
                torch.cuda.synchronize()
                t = time.perf_counter()
                hot_path_step(wl)
                torch.cuda.synchronize()
                if i >= 3:
                    per.append(time.perf_counter() - t)
                    allocs.append(torch.cuda.memory_stats(device)["num_device_alloc"] - s0)
            worst = max(range(len(per)), key=lambda i: per[i])
            srt = sorted(per)
            dt = srt[len(srt) // 2]
            hot = {"what": "the custom ops of one R3Det.simple_test step, alone, same shapes (N=4, C=256, channels_last): "
                           "FeatureRefineModule tail x5 levels (fr_module_nhwc) + refine-head pool x5 levels "
                           "(r3det_levels_pool, 5344 rows / image) + batched multiclass_nms_rotated (v1)",
                   "ms_per_step": round(dt * 1e3, 3), "img_s": round(BATCH / dt, 1),
                   "ms_per_step_mean": round(sum(per) / len(per) * 1e3, 3), "steps": len(per),
                   "slowest_step": {"index": worst, "ms": round(per[worst] * 1e3, 3), "device_allocs_in_it": allocs[worst]},
                   "measured": "before the model is built",
                   "ms_per_step_spread": spread_stats(per)}
            # the same step without the host in it: padded NMS result (no count read), wall and device time
            hot_path_step_sync_free(wl)
            sf = []
            for i in range(3 + 15):
                torch.cuda.synchronize()
                t = time.perf_counter()
                hot_path_step_sync_free(wl)
                torch.cuda.synchronize()
                if i >= 3:
                    sf.append(time.perf_counter() - t)
            hot["sync_free"] = {"what": "the same calls with r3det_mcnms_padded (PaddedNms): nothing reads a count; wall = one "
                                        "step, launched and waited for; device = GPU time per step with the stream kept busy "
                                        "while the host enqueues 10 steps (HIP events) -- what the step costs inside the "
                                        "whole-step graph",
                                "ms_per_step_wall": spread_stats(sf),
                                "ms_per_step_device": round(device_time_ms(lambda: hot_path_step_sync_free(wl)), 3)}
            # ... and as what it is inside GraphedStep: ONE HIP graph, replayed and waited for (wall clock per replay)
            try:
                side = torch.cuda.Stream(device=device)
                side.wait_stream(torch.cuda.current_stream(device))
                with torch.cuda.stream(side):
                    hot_path_step_sync_free(wl)
                torch.cuda.current_stream(device).wait_stream(side)
                torch.cuda.synchronize(device)
                hg = torch.cuda.CUDAGraph()
                with torch.cuda.graph(hg, capture_error_mode="thread_local"):
                    hot_path_step_sync_free(wl)
                sg = []
                for i in range(3 + 15):
                    torch.cuda.synchronize()
                    t = time.perf_counter()
                    hg.replay()
                    torch.cuda.synchronize()
                    if i >= 3:
                        sg.append(time.perf_counter() - t)
                hot["sync_free"]["ms_per_step_graph"] = spread_stats(sg)
                hot["sync_free"]["graph_what"] = ("the same calls captured once in a HIP graph (torch.cuda.CUDAGraph), one replay "
                                                  "launched and waited for per figure: the form the step has inside GraphedStep")
                del hg
            except Exception as e:  # (a capture failure must not cost the bench line)
                hot["sync_free"]["ms_per_step_graph"] = None
                hot["sync_free"]["graph_error"] = repr(e)[:200]
            # the roofline kernel alone, rotating over three buffer sets (3 x 268 MB): every launch reads and
            # writes lines that are NOT in the 256 MiB Infinity Cache -> an HBM figure
            from r3det.ops.feature_refine import fr_module_nhwc
            cl = torch.channels_last
            lv0 = wl["levels"][0]
            sets = [tuple(torch.randn(lv0["a"].shape, device=device).contiguous(memory_format=cl) for _ in range(4))
                    for _ in range(3)]
            state = [0]

            def rot():
                a, b, r, o = sets[state[0] % 3]
                state[0] += 1
                fr_module_nhwc(a, b, wl["bias"], wl["bias"], r, lv0["boxes"], 1.0 / 8, 1, o)
            _C.fr_profile_read()
            _C.set_option("fr_profile", 2)
            timeit(rot, 20, warm=3)
            _C.set_option("fr_profile", 0)
            alone = [r for r in _C.fr_profile_read() if r[0] == BATCH and r[1] == 128]
            if alone:
                us = sum(r[4] for r in alone) / len(alone)
                alone_rec = {"avg_launch_us": round(us, 2), "achieved": round(alg_bytes / us / 1e3, 1),
                             "frac": round(alg_bytes / us / 1e3 / HBM_PEAK_GBS, 4), "launches_timed": len(alone)}
                # The same launch on other box fields: how far the boxes sample from their own cell decides how many tap
                # rows leave the workgroup's regions (DESIGN_HISTORY 4.3 item 7) -- the headline field is the survey's (centres
                # jittered by 0.4 cells); "trained": every 4 x 4 block of positions regresses to one centre, what a
                # trained detector produces around objects; sigma 4: the random-weight bench model's own stage-1 boxes
                by_field = {"survey_sigma_0.4_cells": {"avg_launch_us": alone_rec["avg_launch_us"], "frac": alone_rec["frac"]}}
                base = lv0["boxes"]
                st = 8.0
                trained = base.clone()
                g = (base[:, :2] / (4 * st)).floor() * (4 * st) + 2 * st
                trained[:, :2] = g + torch.randn_like(g) * 0.3 * st
                far = base.clone()
                far[:, :2] = base[:, :2] + torch.randn_like(g) * 4 * st
                for name, bx in (("trained_piles_4x4", trained), ("sigma_4_cells", far)):
                    def rot2():
                        a, b, r, o = sets[state[0] % 3]
                        state[0] += 1
                        fr_module_nhwc(a, b, wl["bias"], wl["bias"], r, bx, 1.0 / 8, 1, o)
                    _C.fr_profile_read()
                    _C.set_option("fr_profile", 2)
                    timeit(rot2, 10, warm=2)
                    _C.set_option("fr_profile", 0)
                    rec = [r for r in _C.fr_profile_read() if r[0] == BATCH and r[1] == 128]
                    if rec:
                        u2 = sum(r[4] for r in rec) / len(rec)
                        by_field[name] = {"avg_launch_us": round(u2, 2), "frac": round(alg_bytes / u2 / 1e3 / HBM_PEAK_GBS, 4)}
                alone_rec["by_field"] = by_field
            del sets, wl
            if not args.no_ops:
                phase("op rates")
                ops = op_rates(device)
            import gc
            gc.collect()
            torch.cuda.empty_cache()

    phase("infer: build + calibrate")
    model, img = build_model(device, seed=100 + rank)
    eager = os.environ.get("R3DET_BENCH_EAGER", "0") == "1"
    phase("infer: capture + warm-up + timed steps")
    whole = None if eager else WholeStep(model, img)
    step = (lambda: model_step(model, img)) if eager else whole
    elapsed, mine, last = timed_region(step, args, device, di)
    ranks = per_rank_ms(mine, args.steps, device, world)
    counts = last.tolist() if eager else whole.counts()
    # ---- beside the headline (rank 0's own clock, a few steps each): the eager step (every launch from Python, the
    # kept counts read by the host, lists packed again: rounds 1-4's timed step) -- its FR launches carry events: the
    # in-model duration of the roofline kernel -- and the step followed by what the reference's timed call ends in
    side = argparse.Namespace(steps=min(args.steps, 10), warmup=2)
    _C.fr_profile_read()                # empty the ring
    _C.set_option("fr_profile", 2)      # the sampler launches of these steps carry a start / stop event
    e_eager, _, _ = timed_region(lambda: model_step(model, img), side, device, di)
    _C.set_option("fr_profile", 0)
    recs = [r for r in _C.fr_profile_read() if r[0] == BATCH and r[1] == 128]
    ms_eager = e_eager / side.steps * 1e3
    ms_host = None
    if whole is not None:
        e_host, _, _ = timed_region(lambda: host_results(whole()), side, device, di)
        ms_host = e_host / side.steps * 1e3

    if rank == 0:
        # Events attached to the launch itself (hipExtLaunchKernelGGL), not stream events around the call:
        # those also time the host's launch gaps whenever the GPU runs ahead of the queue.  The pair reads
        # ~4 us longer than rocprofv3's duration of the same kernel (profiles/: same command under the
        # profiler), so `achieved` errs on the low side.
        span_us = sum(r[4] for r in recs) / max(1, len(recs))
        in_model = alg_bytes / (span_us * 1e-6) / 1e9 if recs else 0.0
        traffic, traffic_src = load_traffic()
        # roofline.achieved / frac: the launch on rotating buffers beyond the Infinity Cache (an HBM figure, the same
        # measurement profiles/r03_fr_nhwc_kernel_stats.txt holds under rocprofv3); the launch inside the timed model
        # steps reads what the convolutions just wrote -- partly from the 256 MiB Infinity Cache -- and is reported
        # next to it as in_model_l3_assisted.  With --model-only only the in-model figure exists.
        head = alone_rec or {"avg_launch_us": round(span_us, 2), "achieved": round(in_model, 1),
                             "frac": round(in_model / HBM_PEAK_GBS, 4), "launches_timed": len(recs)}
        line = dict(common, **{
            "metric": "img/s, R3Det R50-FPN 1024x1024 inference (r3det_r50_fpn_1x v1)",
            "value": round(world * BATCH * args.steps / elapsed, 2),
            "unit": "img/s",
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "per_rank_ms_per_step": ranks, **dist_info(),
            "config": {"workload_short": "BASELINE configs[2]: r3det_r50_fpn_1x v1 + FeatureRefineModule, batch 4 x 1024x1024 "
                                         "per GPU, full inference, synthetic",
                       "workload": "BASELINE configs[2] (the single-GPU case of the metric's model): "
                                   "r3det_r50_fpn_1x v1 + FeatureRefineModule, batch=4 x 1024x1024 per GPU, "
                                   "full inference incl. backbone, random-init weights, score bias calibrated "
                                   "to ~1 % candidates",
                       "batch_per_gpu": BATCH, "global_batch": BATCH * world, "nms_type": "v1",
                       "parallelism": f"image-parallel x{world}, all_gather of detections"},
            "roofline": {"bound": "hbm",
                         "kernel": "fr_forward_nhwc_wide<true> = the FeatureRefineModule tail at level 0 "
                                   "(4x256x128x128, channels_last): (conv_a + bias) + (conv_b + bias), sampler, residual in "
                                   "one launch, 3 reads + 1 write per element; duration = the launch's own start/stop HIP "
                                   "events (hipExtLaunchKernelGGL)",
                         "measured_on": ("the launch alone over 3 rotating buffer sets (0.8 GB: beyond the 256 MiB "
                                         "Infinity Cache)") if alone_rec else "inside the timed model steps (--model-only)",
                         "achieved": head["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": head["frac"],
                         "avg_launch_us": head["avg_launch_us"], "launches_timed": head["launches_timed"],
                         "traffic": traffic, "traffic_source": traffic_src, "alg_bytes_per_launch": alg_bytes,
                         "sampler_only_frac": round(head["frac"] * (2 * 4 * BATCH * C * H * W + 20 * BATCH * H * W)
                                                    / alg_bytes, 4),
                         "sampler_only_note": "the same launch priced at SURVEY 8d's B_fr (1 read + 1 write per element: "
                                              "135.5 MB) instead of the 16 B / element the fused launch has to move",
                         "by_field": dict((alone_rec or {}).get("by_field", {}),
                                          in_model={"avg_launch_us": round(span_us, 2),
                                                    "frac": round(in_model / HBM_PEAK_GBS, 4),
                                                    "note": "the random-weight model's own stage-1 boxes (offsets of "
                                                            "4.7 +- 1.9 cells), inputs Infinity-Cache warm"}),
                         "by_field_note": "frac is a property of the launch AND of how far the boxes sample from their own "
                                          "cell; `frac` above is the survey's field (centres jittered by 0.4 cells)",
                         "in_model_l3_assisted": {"avg_launch_us": round(span_us, 2), "achieved": round(in_model, 1),
                                                  "frac": round(in_model / HBM_PEAK_GBS, 4), "launches_timed": len(recs),
                                                  "what": "the same launch inside the timed model steps: its three "
                                                          "inputs were just written by the convolutions (Infinity "
                                                          "Cache-assisted, not an HBM fraction)"}},
            "kept_per_image": [int(c) for c in counts],
            "timed_step": {
                "what": ("R3Det.simple_test as ONE HIP graph per step (backbone, neck, heads, FRM, decoding, per-level "
                         "pool, batched multiclass NMS, padded [B, 2000 + 1, 7] result) + the step's one exchange "
                         "(gather_padded); no host synchronisation inside, detections stay on the device")
                if not eager else "eager step (R3DET_BENCH_EAGER=1): every launch from Python, kept counts read by the host",
                "rbbox2result_inside": False,
                "note": "the reference's timed call (tools/analysis_tools/benchmark.py:104-110) ends in rbbox2result "
                        "(models/detectors/r3det.py:137-143: detections to the host, per-class numpy lists); "
                        "ms_per_step_with_host_results is the same step followed by exactly that",
                "ms_per_step_with_host_results": None if ms_host is None else round(ms_host, 3),
                "img_s_with_host_results": None if ms_host is None else round(world * BATCH / ms_host * 1e3, 2),
                "ms_per_step_eager": round(ms_eager, 3),
                "steps_of_the_side_figures": side.steps,
                "capacity_redos": 0 if whole is None else whole.redone,
                "candidate_capacity": None if whole is None else whole.g.nms.cap},
        })
        if whole is not None and not args.model_only:
            with section("infer: the two candidate pools"):

                def pool_record(ws, ms):
                    boxes, scores = ws.g.model.dense_test(ws.g.static_in)
                    boxes, scores = boxes.contiguous(), scores.contiguous()
                    per_class = (scores[..., :-1] > SCORE_THR).sum((0, 1)).float()
                    nms_ms = device_time_ms(lambda: ws.g.nms(boxes, scores))
                    return {"ms_per_step": round(ms, 3), "img_s": round(BATCH / ms * 1e3, 2),
                            "candidates_per_image": int(per_class.sum().item() / BATCH),
                            "largest_class_share": round(float(per_class.max() / per_class.sum().clamp(min=1)), 3),
                            "kept_per_image": ws.counts(),
                            "nms_device_us": round(nms_ms * 1e3, 1)}
                by_pool = {"what": "the same step on the two calibrations of the random-weight model's score bias; "
                                   "nms_device_us = r3det_mcnms_select + r3det_mcnms_padded on the step's own pool, GPU time "
                                   "with the queue kept full (no host in it)",
                           "spread_15_classes": dict(pool_record(whole, elapsed / args.steps * 1e3),
                                                     note="the headline: one bias shift per class, equal shares (SURVEY 8d)")}
                if world == 1:
                    m1, i1 = build_model(device, seed=100 + rank, spread=False)
                    w1 = WholeStep(m1, i1)
                    e1, _, _ = timed_region(w1, side, device, di)
                    by_pool["one_label"] = dict(pool_record(w1, e1 / side.steps * 1e3),
                                                note="rounds 1-4's pool: ONE common shift (bounded: "
                                                     f"{side.steps} steps)")
                    del m1, i1, w1
                line["by_pool"] = by_pool
        if hot is not None:
            line["hot_path"] = hot
        if ops is not None:
            line["ops"] = ops
        del model, img
        import gc
        gc.collect()              # the module graph has reference cycles: free it now
        torch.cuda.empty_cache()
        if world == 1 and not args.no_extras:
            torch.cuda.empty_cache()
            ex = argparse.Namespace(steps=5, warmup=3)
            with section("extra: rretinanet (configs[1])"):
                m2, i2 = build_model(device, 200, "RRetinaNet", RRETINA_BATCH)
                w2 = WholeStep(m2, i2)
                e2, _, _ = timed_region(w2, ex, device, di)
                c2 = torch.tensor(w2.counts())
                ee2, _, _ = timed_region(lambda: model_step(m2, i2), ex, device, di)
                line["rretinanet"] = {"workload": "BASELINE configs[1]: rretinanet_obb_r50_fpn v1, batch=2 x 1024x1024, "
                                                  "inference, 8576-box pools per image, nms v1 (bounded: 5 steps; --mode "
                                                  "rretinanet times it as the main region)",
                                      "img_s": round(RRETINA_BATCH * ex.steps / e2, 2),
                                      "ms_per_step": round(e2 / ex.steps * 1e3, 3),
                                      "ms_per_step_eager": round(ee2 / ex.steps * 1e3, 3),
                                      "timed_step": "the whole step as one HIP graph + gather_padded (as the headline)",
                                      "kept_per_image": [int(c) for c in c2.tolist()]}
                del m2, i2, w2
                torch.cuda.empty_cache()
            with section("extra: train (configs[4])"):
                tr = build_train(device, 300, 1)
                e3, _, loss = timed_region(lambda: train_step(tr), ex, device, di)
                ta, tf = train_custom_op_ms(tr, device)
                ms3 = e3 / ex.steps * 1e3
                line["train"] = {"workload": "BASELINE configs[4] on one GPU: r3det_r50_fpn_1x v1 training step, batch=2 x "
                                             "1024x1024, 128 GT per image (bounded: 5 steps; --mode train times it as the main "
                                             "region, with DDP at N > 1)",
                                 "img_s": round(TRAIN_BATCH * ex.steps / e3, 2), "ms_per_step": round(ms3, 3),
                                 "final_loss": round(float(loss), 4),
                                 "layout": "channels_last" if TRAIN_CHANNELS_LAST else "NCHW",
                                 "custom_ops_isolated": {"assign_ms": round(ta, 3), "fr_fwd_bwd_ms": round(tf, 3),
                                                         "share_of_step": round((ta + tf) / ms3, 4),
                                                         "detail": getattr(train_custom_op_ms, "detail", None)}}
                del tr
                torch.cuda.empty_cache()
                # the same step in the other layout (3 steps): channels_last runs the FR sampler + backward on NHWC memory
                tr = build_train(device, 300, 1, channels_last=not TRAIN_CHANNELS_LAST)
                ex3 = argparse.Namespace(steps=3, warmup=2)
                e4, _, _ = timed_region(lambda: train_step(tr), ex3, device, di)
                _, tf4 = train_custom_op_ms(tr, device)
                line["train"]["other_layout"] = {"layout": "NCHW" if TRAIN_CHANNELS_LAST else "channels_last",
                                                 "ms_per_step": round(e4 / ex3.steps * 1e3, 3),
                                                 "fr_fwd_bwd_ms_isolated": round(tf4, 3)}
                del tr
        if world == 1 and not args.no_cpu_baseline:
            with section("cpu baseline"):
                line["cpu_baseline"] = cpu_baseline()
        phase("done")
        emit(line)
    if world > 1:
        di.barrier(device)
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
