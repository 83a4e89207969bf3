"""batched_rnms (v1) on the bench pools n = 2000 / 5344 / 8576 and the batched multiclass pipeline on
4 x 5344-box pools, for rocprofv3 --kernel-trace (profiles/*_nms_*).  NMS_PROF_N=8576 runs one size only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from r3det import synthetic as syn  # noqa: E402
from r3det.core.post_processing import CapacityHint, multiclass_nms_rotated_batch  # noqa: E402
from r3det.ops import batched_rnms  # noqa: E402

dev = torch.device("cuda")
from r3det.ops import obb_batched_nms  # noqa: E402
from r3det import _C  # noqa: E402

if os.environ.get("NMS_PROF_clip_impl"):  # A/B: 1 = the LDS-list clip of rounds 2-4
    _C.set_option("clip_impl", int(os.environ["NMS_PROF_clip_impl"]))
if os.environ.get("NMS_PROF_nms_impl"):   # A/B: 4 = the walk reducer
    _C.set_option("nms_impl", int(os.environ["NMS_PROF_nms_impl"]))

if os.environ.get("NMS_PROF_FAST_MAX_N"):  # A/B: the one-call form's crossover (r3det/ops/nms.py)
    import r3det.ops.nms as _nms
    _nms.FAST_MAX_N = int(os.environ["NMS_PROF_FAST_MAX_N"])
sizes = [os.environ["NMS_PROF_N"]] if os.environ.get("NMS_PROF_N") else [2000, 5344, 8576]
for n in sizes:
    if str(n).startswith("v3_"):  # the v3 family (nms_rotated_ext.nms_rotated behind obb_batched_nms)
        n, batched_rnms = int(str(n)[3:]), obb_batched_nms
    n = int(n)
    mb, ms = syn.nms_pool(n * 10 // 6 + 64, 77 + n, device=dev)
    sc, lab = ms[:, :-1].max(1)
    idx = torch.nonzero(sc > 0.05).squeeze(1)[:n]
    b, s, l = mb[idx].contiguous(), sc[idx].contiguous(), lab[idx].contiguous()
    if os.environ.get("DUMP_POOL"):  # for tools/probes/nms_reduce_probe.hip: n | boxes | scores | labels, all f32
        import numpy as np
        with open(os.environ["DUMP_POOL"], "wb") as f:
            np.array([b.size(0)], np.float32).tofile(f)
            b.cpu().numpy().astype(np.float32).tofile(f)
            s.cpu().numpy().astype(np.float32).tofile(f)
            l.cpu().numpy().astype(np.float32).tofile(f)
    for _ in range(3):
        batched_rnms(b, s, l, 0.1)
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(20):
        d, k = batched_rnms(b, s, l, 0.1)
    en.record()
    torch.cuda.synchronize()
    print(f"batched_rnms n={b.size(0)} kept={k.numel()}: {st.elapsed_time(en) * 50:8.1f} us per call", flush=True)
if not os.environ.get("NMS_PROF_N"):
    pools = [syn.nms_pool(syn.R3DET_POOL, 7000 + i, device=dev) for i in range(4)]
    pb, ps = torch.stack([p[0] for p in pools]), torch.stack([p[1] for p in pools])
    hint = CapacityHint()
    for _ in range(3):
        multiclass_nms_rotated_batch(pb, ps, 0.05, dict(iou_thr=0.1), 2000, hint=hint)
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(20):
        res = multiclass_nms_rotated_batch(pb, ps, 0.05, dict(iou_thr=0.1), 2000, hint=hint)
    en.record()
    torch.cuda.synchronize()
    print(f"multiclass_nms_rotated_batch 4 x 5344: {st.elapsed_time(en) * 50:8.1f} us per call, kept "
          f"{[r[0].size(0) for r in res]}", flush=True)
