"""Image-parallel inference plumbing: one process per GPU, tiles sharded round-robin, and ONE
exchange step -- gathering the final detections on rank 0.  The hot-path ops
themselves never communicate.  Works with backend 'nccl' (= RCCL over xGMI on ROCm) on GPU
tensors and with 'gloo' on CPU tensors (used by the world_size-2 CPU tests).
"""
import os

import torch
import torch.distributed as dist

MAX_PER_IMG = 2000  # test_cfg.max_per_img (configs/r3det/r3det_r50_fpn_1x_dota_v1.py:104)
DET_COLS = 7        # cx, cy, w, h, theta, score, label


def env_world():
    """(rank, local_rank, world_size) from the torch.distributed.run environment."""
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)),
            int(os.environ.get("WORLD_SIZE", 1)))


def force_group():
    """R3DET_FORCE_DIST=1: build the process group (and the DDP wrapper) even for ONE rank, so that the RCCL code
    path -- init with device_id, all_gather_into_tensor, bucketed gradient all-reduce -- can be run on a single-GPU
    box (tests/test_gpu_dist_single.py)."""
    return os.environ.get("R3DET_FORCE_DIST", "0") == "1"


def init(backend=None, device=None):
    rank, local_rank, world = env_world()
    if (world > 1 or (force_group() and "MASTER_ADDR" in os.environ)) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def shard_tiles(num_tiles, rank, world):
    """Tile indices of this rank: r, r + world, r + 2 world, ... (round-robin, SURVEY 8d cfg 4)."""
    return list(range(rank, num_tiles, world))


def pack_detections(dets_list, labels_list, max_per_img=MAX_PER_IMG):
    """[(k_i, 6)], [(k_i,)] -> padded (B, max_per_img, 7) fp32 + counts (B,) int32."""
    B = len(dets_list)
    ref = dets_list[0] if B else torch.zeros(0, 6)
    out = ref.new_zeros((B, max_per_img, DET_COLS), dtype=torch.float32)
    counts = torch.zeros(B, dtype=torch.int32, device=ref.device)
    for i, (d, l) in enumerate(zip(dets_list, labels_list)):
        k = min(d.size(0), max_per_img)
        out[i, :k, :6] = d[:k]
        out[i, :k, 6] = l[:k].to(torch.float32)
        counts[i] = k
    return out, counts


def gather_detections(packed, counts, dst=0, batch_size=None):
    """Gather every rank's padded detections on ``dst``.  Returns (list_of_packed, list_of_counts) on dst -- one
    entry per rank, each ``batch_size`` images long with count -1 for the images a rank did not have
    (``unpack_detections`` skips them) -- and (None, None) elsewhere; single-process: passthrough.

    ONE collective per call, the same for RCCL ('nccl') and gloo: ``all_gather_into_tensor`` of the detections with
    the counts riding as one more row per image (fp32 holds them exactly).  It needs identical shapes on every rank,
    so every rank first pads its batch to a common size: ``batch_size`` -- the configured per-rank batch, which the
    inference loop knows; EVERY rank must pass the same value (or all None), otherwise the collectives mismatch and
    hang -- or, with None, the MAX of the ranks' batch sizes (one more 8-byte all-reduce and a host read).  A rank that
    has run out of images still has to take part: it calls with an empty batch (B = 0).  Count -1 marks the padding;
    an uneven last batch therefore neither hangs nor mixes up ranks.
    Needs torch >= 1.13 (``all_gather_into_tensor`` on both backends)."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force_group()):
        return [packed], [counts]
    world, rank = dist.get_world_size(), dist.get_rank()
    B = packed.size(0)
    if packed.dtype != torch.float32:  # the counts ride in the detections' tensor: exact up to 2^24 in fp32 only
        raise TypeError(f"gather_detections needs fp32 detections, got {packed.dtype}")
    if batch_size is None:
        m = torch.tensor([B], dtype=torch.int64, device=packed.device)
        dist.all_reduce(m, op=dist.ReduceOp.MAX)
        batch_size = int(m.item())
    if B > batch_size:
        raise RuntimeError(f"rank {rank}: batch of {B} images exceeds the common batch size {batch_size}")
    # [batch_size, rows + 1, 7]: the images' detections, then one row whose first value is the count (-1: padding)
    rows, cols = packed.size(1), packed.size(2)
    mine = packed.new_zeros((batch_size, rows + 1, cols))
    mine[:B, :rows] = packed
    mine[:, rows, 0] = -1.0
    mine[:B, rows, 0] = counts.to(packed.dtype)
    everyone = packed.new_empty((world * batch_size, rows + 1, cols))
    # (RCCL has no native gather-to-one that beats all_gather at 56 KB / image)
    dist.all_gather_into_tensor(everyone, mine)
    if rank != dst:
        return None, None
    # (no host read here: the padding is marked in the counts)
    allc = everyone[:, rows, 0].to(counts.dtype)
    return [p[:, :rows] for p in everyone.split(batch_size)], list(allc.split(batch_size))


def gather_padded(mine, dst=0):
    """The step's ONE exchange on the buffer the detector already laid out (``PaddedNms.out`` /
    ``GraphedStep``): (batch_size, rows + 1, 7) fp32, row ``rows`` of every image = [count, 0, ...] (-1: an image this
    rank did not have).  No packing pass, no host read; every rank passes the same shape.  Returns the gathered
    (world * batch_size, rows + 1, 7) tensor on ``dst`` (``split_gathered`` turns it into gather_detections' lists),
    None elsewhere; single process: ``mine`` itself."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force_group()):
        return mine
    if mine.dtype != torch.float32:
        raise TypeError(f"gather_padded needs fp32 detections, got {mine.dtype}")
    world, rank = dist.get_world_size(), dist.get_rank()
    everyone = mine.new_empty((world * mine.size(0),) + tuple(mine.shape[1:]))
    dist.all_gather_into_tensor(everyone, mine)
    return everyone if rank == dst else None


def split_gathered(everyone, batch_size):
    """gather_padded's tensor -> gather_detections' (list_of_packed, list_of_counts), one entry per rank."""
    rows = everyone.size(1) - 1
    allc = everyone[:, rows, 0].to(torch.int32)
    return [p[:, :rows] for p in everyone.split(batch_size)], list(allc.split(batch_size))


def unpack_detections(packed, counts):
    """Inverse of pack_detections for one rank's tensors -> [(dets (k,6), labels (k,))]; entries with count -1 (the
    padding ``gather_detections`` adds behind a short batch) are skipped."""
    out = []
    for i, k in enumerate(counts.tolist()):  # (one host read for the whole batch)
        if k >= 0:
            out.append((packed[i, :k, :6], packed[i, :k, 6].to(torch.long)))
    return out


def max_over_ranks(seconds, device):
    """Timing convention of bench.py: the slowest rank defines the step time."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force_group()):
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier(device=None):
    if dist.is_initialized() and (dist.get_world_size() > 1 or force_group()):
        if dist.get_backend() == "nccl" and device is not None:
            dist.barrier(device_ids=[device.index])
        else:
            dist.barrier()
