"""Worker of tests/test_gpu_dist_single.py: ONE rank under torch.distributed.run with R3DET_FORCE_DIST=1, so that
the RCCL path of the multi-GPU runs (process group with device_id, all_gather_into_tensor of the packed
detections, all-reduce of the timing, DDP's bucketed gradient all-reduce) executes on a single-GPU box."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from r3det import dist_infer as di  # noqa: E402
from r3det import dist_train as dt  # noqa: E402
from r3det.models import R3Det  # noqa: E402

rank, local_rank, world = di.env_world()
device = torch.device("cuda", local_rank)
torch.cuda.set_device(device)
di.init(device=device)
assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1
# inference exchange
dets = [torch.rand(5, 6, device=device), torch.rand(0, 6, device=device)]
labs = [torch.randint(0, 15, (5,), device=device), torch.zeros(0, dtype=torch.long, device=device)]
packed, counts = di.pack_detections(dets, labs)
gp, gc = di.gather_detections(packed, counts)
assert len(gp) == 1 and torch.equal(gp[0], packed) and gc[0].tolist() == [5, 0]
assert abs(di.max_over_ranks(1.25, device) - 1.25) < 1e-12
di.barrier(device)
# round 5: the whole inference step as one HIP graph, recorded while the process group is alive, and the step's exchange
# on the buffer the graph wrote (gather_padded -> all_gather_into_tensor over RCCL)
from r3det.models.detectors import GraphedStep, calibrate_score_bias  # noqa: E402
torch.manual_seed(5)
infer = R3Det().eval().to(device)
im = torch.randn(2, 3, 256, 256, device=device)
calibrate_score_bias(infer, im, frac=0.02, per_class=True)
gstep = GraphedStep(infer, im)
for _ in range(3):
    out, redo = gstep.step(im)
    everyone = di.gather_padded(out)
    assert not redo and everyone.shape == out.shape
torch.cuda.synchronize()
assert torch.equal(everyone, out)
gpk, gcn = di.split_gathered(everyone, 2)
assert len(gpk) == 1 and gcn[0].tolist() == [int(c) for c in out[:, -1, 0].tolist()] and min(gcn[0].tolist()) > 0
di.barrier(device)
del infer, gstep
# training: DDP over RCCL, one step on a tiny batch
from test_train_cpu import tiny_batch  # noqa: E402
torch.manual_seed(7)
model = R3Det().to(device).train()
ddp = dt.wrap_ddp(model, device=device, bucket_cap_mb=4)
assert isinstance(ddp, torch.nn.parallel.DistributedDataParallel)
opt = dt.build_optimizer(model, lr=0.001)
img, gtb, gtl = tiny_batch(11, n=2, n_gt=6)
img = img.to(device)
gtb = [g.to(device) for g in gtb]
gtl = [g.to(device) for g in gtl]
loss, log_vars = dt.train_step(ddp, opt, img, gtb, gtl)
torch.cuda.synchronize()
assert torch.isfinite(loss)
print(json.dumps({"backend": dist.get_backend(), "world": dist.get_world_size(), "loss": float(loss)}))
di.barrier(device)
dist.destroy_process_group()
