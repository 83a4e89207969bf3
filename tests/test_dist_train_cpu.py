"""CPU, world_size 2 over gloo: the data-parallel training step of BASELINE configs[4]
(r3det.dist_train).  Each rank runs R3Det.forward_train on ITS images; DistributedDataParallel
averages the gradients.  Checked against a single process that computes the two ranks' losses one
after the other and backpropagates their mean -- which is what the reference's arrangement computes
(per-rank loss normalisation, see r3det/dist_train.py).  The HIP kernels are replaced by the TEST-ONLY
stand-ins of tests/cpu_standins.py; the assignment, FR forward/backward and loss glue are the product's."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _paths():
    for p in (HERE, ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _data(rank):
    from test_train_cpu import tiny_batch
    return tiny_batch(100 + rank, n=2 if rank == 0 else 1, n_gt=5 + 2 * rank)  # uneven batches and GT counts


def _model():
    from r3det.models import R3Det
    torch.manual_seed(1234)
    return R3Det().train()


def _worker(rank, world, port, out_path):
    _paths()
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from cpu_standins import cpu_kernels
    from r3det import dist_infer as di
    from r3det import dist_train as dt
    di.init(backend="gloo")
    model = _model()
    ddp = dt.wrap_ddp(model, bucket_cap_mb=4)  # several buckets even for this test's gradient sizes
    assert isinstance(ddp, torch.nn.parallel.DistributedDataParallel)
    opt = dt.build_optimizer(model, lr=0.01, momentum=0.0, weight_decay=0.0)
    img, gtb, gtl = _data(rank)
    with cpu_kernels():
        loss, log_vars = dt.train_step(ddp, opt, img, gtb, gtl)
    grads = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    # every rank holds the same averaged gradient and the same updated weights
    flat = torch.cat([g.reshape(-1) for g in grads.values()])
    both = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    assert torch.equal(both[0], both[1])
    if rank == 0:
        torch.save(dict(grads=grads, loss=float(loss), keys=sorted(log_vars),
                        w=model.refine_head[0].retina_cls.weight.detach().clone()), out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_gradients_equal_mean_of_per_rank_gradients(tmp_path):
    _paths()
    out = str(tmp_path / "ddp.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    assert got["keys"] == ['loss', 's0.loss_bbox', 's0.loss_cls', 'sr0.loss_bbox', 'sr0.loss_cls']

    from cpu_standins import cpu_kernels
    from r3det.models.detectors import parse_losses
    model = _model()
    w0 = model.refine_head[0].retina_cls.weight.detach().clone()
    total = 0
    with cpu_kernels():
        for rank in range(2):
            img, gtb, gtl = _data(rank)
            loss, _ = parse_losses(model(img, return_loss=True, gt_bboxes=gtb, gt_labels=gtl))
            total = total + loss / 2
        total.backward()
    want = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    assert sorted(want) == sorted(got["grads"])
    for n, g in want.items():
        scale = float(g.abs().max()) + 1e-12
        assert float((got["grads"][n] - g).abs().max()) <= 1e-5 * scale + 1e-9, n
    # one SGD step with lr 0.01 moved the weights by -lr * averaged gradient
    step = w0 - 0.01 * want['refine_head.0.retina_cls.weight']
    assert torch.allclose(got["w"], step, rtol=1e-5, atol=1e-8)


def test_wrap_ddp_single_process_is_identity():
    _paths()
    from r3det import dist_train as dt
    m = torch.nn.Linear(2, 2)
    assert dt.wrap_ddp(m) is m
