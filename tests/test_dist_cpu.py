"""CPU, world_size 2 over gloo: the image-parallel plumbing of r3det.dist_infer -- tile
sharding, packing, the single exchange step (gather of detections) and the max-over-ranks
timing convention that bench.py uses on RCCL."""
import os
import socket
import sys

import pytest
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


def _fake_detections(tile):
    """Deterministic per-tile 'detections' so rank 0 can verify what every rank produced."""
    g = torch.Generator().manual_seed(1000 + tile)
    k = 5 + (tile * 7) % 40
    dets = torch.rand(k, 6, generator=g)
    labels = torch.randint(0, 15, (k,), generator=g)
    return dets, labels


def _worker(rank, world, port, num_tiles, batch, out_path):
    for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from r3det import dist_infer as di
    r, lr, w = di.init(backend="gloo")
    assert (r, w) == (rank, world)
    mine = di.shard_tiles(num_tiles, rank, world)
    collected = {}
    most = max(len(di.shard_tiles(num_tiles, r_, world)) for r_ in range(world))
    for b0 in range(0, most, batch):  # (every rank takes part in every round: one that has run out calls with B = 0)
        tiles = mine[b0:b0 + batch]
        dl = [_fake_detections(t) for t in tiles]
        if tiles:
            packed, counts = di.pack_detections([d for d, _ in dl], [l for _, l in dl], max_per_img=64)
        else:
            packed, counts = torch.zeros(0, 64, 7), torch.zeros(0, dtype=torch.int32)
        # a rank's last batch may be short (13 tiles over 2 ranks, batch 4: 4 + 3 and 4 + 2): gather_detections
        # pads it itself -- to the configured batch size in the even rounds, to the MAX over the ranks (one more
        # tiny all-reduce) in the odd ones -- and all ranks go through the same all_gather_into_tensor
        gp, gc = di.gather_detections(packed, counts, dst=0, batch_size=batch if (b0 // batch) % 2 == 0 else None)
        if rank == 0:
            for src in range(world):
                src_tiles = di.shard_tiles(num_tiles, src, world)[b0:b0 + batch]
                got = di.unpack_detections(gp[src], gc[src])
                assert len(got) == len(src_tiles) and gp[src].size(0) >= len(src_tiles)
                for j, (d, l) in enumerate(got):
                    collected[src_tiles[j]] = (d.clone(), l.clone())
        else:
            assert gp is None and gc is None
    t = di.max_over_ranks(0.25 + rank, torch.device("cpu"))
    assert t == pytest.approx(0.25 + world - 1)
    di.barrier()
    if rank == 0:
        torch.save(collected, out_path)
    dist.destroy_process_group()


@pytest.mark.parametrize("num_tiles", [13, 9])  # 9: ranks with 4 + 1 and 4 + 0 tiles -- one rank misses a whole batch
def test_image_parallel_gather_world2(tmp_path, num_tiles):
    world, batch = 2, 4
    out = str(tmp_path / "collected.pt")
    mp.spawn(_worker, args=(world, _free_port(), num_tiles, batch, out), nprocs=world, join=True)
    got = torch.load(out)
    assert sorted(got.keys()) == list(range(num_tiles))  # every tile exactly once
    for t in range(num_tiles):
        d, l = _fake_detections(t)
        assert torch.equal(got[t][0], d) and torch.equal(got[t][1], l)


def test_shard_and_pack_single_process():
    sys.path.insert(0, os.path.join(ROOT, "r3det-pytorch_amd"))
    from r3det import dist_infer as di
    assert di.shard_tiles(10, 1, 4) == [1, 5, 9]
    assert sum(len(di.shard_tiles(1024, r, 8)) for r in range(8)) == 1024
    dets = [torch.rand(3, 6), torch.rand(0, 6), torch.rand(70, 6)]
    labels = [torch.tensor([1, 2, 3]), torch.zeros(0, dtype=torch.long), torch.arange(70) % 15]
    packed, counts = di.pack_detections(dets, labels, max_per_img=64)
    assert packed.shape == (3, 64, 7) and counts.tolist() == [3, 0, 64]  # truncated at max_per_img
    back = di.unpack_detections(packed, counts)
    assert torch.equal(back[0][0], dets[0]) and torch.equal(back[0][1], labels[0])
    assert back[1][0].shape == (0, 6)
    assert torch.equal(back[2][0], dets[2][:64])
    gp, gc = di.gather_detections(packed, counts)  # no process group: passthrough
    assert gp[0] is packed and gc[0] is counts


class _StubModel:
    """simple_test of a detector: per-image (dets (k, 6), labels (k,)); rank 1's second image is empty and
    its batch is one image short (uneven last batch)."""

    def __init__(self, rank):
        self.rank = rank

    def simple_test(self, img):
        out = []
        for i in range(img.size(0)):
            k = 0 if (self.rank == 1 and i == 1) else 3 + 2 * i + self.rank
            g = torch.Generator().manual_seed(10 * self.rank + i)
            out.append((torch.rand(k, 6, generator=g), torch.randint(0, 15, (k,), generator=g)))
        return out


def _bench_worker(rank, world, port, out_path):
    for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import argparse

    import bench
    from r3det import dist_infer as di
    di.init(backend="gloo")
    dev = torch.device("cpu")
    model = _StubModel(rank)
    img = torch.zeros(4 if rank == 0 else 3, 3, 8, 8)
    # count the collectives of a step: ONE all_gather_into_tensor (the counts ride in it) and nothing else -- no
    # all-reduce, no host read -- when the configured batch is passed, as bench.py does
    calls = []
    real_ag, real_ar = dist.all_gather_into_tensor, dist.all_reduce
    dist.all_gather_into_tensor = lambda *a, **k: (calls.append("all_gather_into_tensor"), real_ag(*a, **k))[1]
    dist.all_reduce = lambda *a, **k: (calls.append("all_reduce"), real_ar(*a, **k))[1]
    counts = bench.model_step(model, img, batch_size=4)  # pack + the step's one exchange (gather on rank 0)
    dist.all_gather_into_tensor, dist.all_reduce = real_ag, real_ar
    assert calls == ["all_gather_into_tensor"], calls
    assert counts.tolist() == ([3, 5, 7, 9] if rank == 0 else [4, 0, 8])
    elapsed, mine, last = bench.timed_region(lambda: bench.model_step(model, img, batch_size=4),
                                             argparse.Namespace(steps=3, warmup=1), dev, di)
    assert elapsed >= mine - 1e-9 and torch.equal(last, counts)
    ranks = bench.per_rank_ms(mine, 3, dev, world)
    info = bench.dist_info()
    assert len(ranks) == world and info == {"backend": "gloo", "rccl_ranks": 2}
    if rank == 0:
        torch.save(dict(ranks=ranks, info=info, elapsed=elapsed), out_path)
    di.barrier()
    dist.destroy_process_group()


def test_bench_step_gather_and_rank_metadata_world2(tmp_path):
    """bench.py's own model_step / timed_region / per-rank times / backend + rank count under gloo,
    world size 2, with ranks whose batches differ in size (the gather of detections is per rank: shapes
    differ across ranks: the shorter one is padded inside gather_detections, both go through the same
    all_gather_into_tensor the RCCL runs use) and an image without detections."""
    out = str(tmp_path / "bench.pt")
    mp.spawn(_bench_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    assert got["info"]["rccl_ranks"] == 2 and len(got["ranks"]) == 2
    assert got["elapsed"] * 1e3 / 3 >= max(got["ranks"]) - 1e-3  # (the per-rank times are rounded to 1 us)


def _padded_worker(rank, world, port, out_path):
    for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from r3det import dist_infer as di
    di.init(backend="gloo")
    batch, rows = 3, 16
    # the buffer a detector's padded NMS writes (PaddedNms.out / GraphedStep): rows of 7, then the count row;
    # rank 1 has only two images of the batch: the third is marked -1
    mine = torch.zeros(batch, rows + 1, 7)
    want = {}
    for i in range(batch if rank == 0 else 2):
        k = 2 + 3 * i + rank
        g = torch.Generator().manual_seed(50 * rank + i)
        d = torch.rand(k, 6, generator=g)
        lab = torch.randint(0, 15, (k,), generator=g)
        mine[i, :k, :6], mine[i, :k, 6], mine[i, rows, 0] = d, lab.float(), k
        want[i] = (d, lab)
    if rank == 1:
        mine[2, rows, 0] = -1.0
    calls = []
    real_ag, real_ar = dist.all_gather_into_tensor, dist.all_reduce
    dist.all_gather_into_tensor = lambda *a, **k: (calls.append("all_gather_into_tensor"), real_ag(*a, **k))[1]
    dist.all_reduce = lambda *a, **k: (calls.append("all_reduce"), real_ar(*a, **k))[1]
    everyone = di.gather_padded(mine, dst=0)
    dist.all_gather_into_tensor, dist.all_reduce = real_ag, real_ar
    assert calls == ["all_gather_into_tensor"], calls  # the step's ONE exchange, on the buffer as it is
    if rank == 0:
        packed, counts = di.split_gathered(everyone, batch)
        assert [c.tolist() for c in counts] == [[2, 5, 8], [3, 6, -1]]
        got0 = di.unpack_detections(packed[0], counts[0])
        got1 = di.unpack_detections(packed[1], counts[1])
        assert len(got0) == 3 and len(got1) == 2  # (the image rank 1 did not have is skipped)
        for i, (d, lab) in enumerate(got0):
            assert torch.equal(d, want[i][0]) and torch.equal(lab, want[i][1])
        torch.save(dict(n=[len(got0), len(got1)]), out_path)
    else:
        assert everyone is None
    di.barrier()
    dist.destroy_process_group()


def test_gather_padded_world2(tmp_path):
    """Round 5: the exchange on the padded buffer the whole-step graph writes -- one all_gather_into_tensor, no packing
    pass, no host read; an image a rank does not have is marked -1 in its count row."""
    out = str(tmp_path / "padded.pt")
    mp.spawn(_padded_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert torch.load(out)["n"] == [3, 2]
