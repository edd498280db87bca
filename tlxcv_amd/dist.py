"""Batch-sharded inference across the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, xGMI underneath).  Images
are independent units: the batch is split into contiguous shards, weights are replicated, no
activation ever crosses GPUs, and the only exchange is ONE all-gather of the logits
((B/world, classes) per rank -> (B, classes) everywhere) — 512 KB per rank for ResNet-50 at 256
images: latency-bound on the point-to-point xGMI mesh, so a single un-bucketed collective on the
compute stream is the right shape.  The reference has no counterpart (its all_gather is a stub,
tlxcv/tasks/human_pose_estimation.py:373-374).
"""
import os

import torch
import torch.distributed as dist


def init(backend=None):
    """Initialise from torchrun's env (RANK/WORLD_SIZE/LOCAL_RANK/MASTER_*).  Returns (rank, world, local)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("TLXMI_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            torch.cuda.set_device(local % torch.cuda.device_count())
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_bounds(n, rank, world):
    """Contiguous shard [lo, hi) of n units for `rank`; the first n % world ranks get one extra."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(x, rank=None, world=None):
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
        world = dist.get_world_size() if dist.is_initialized() else 1
    lo, hi = shard_bounds(x.shape[0], rank, world)
    return x[lo:hi]


def all_gather_logits(local_logits, total=None, group=None):
    """(b_r, classes) on every rank -> (sum b_r, classes) on every rank, rank order = batch order.
    Equal shards use one all_gather_into_tensor; ragged shards pad to the largest shard."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local_logits
    world = dist.get_world_size(group)
    local_logits = local_logits.contiguous()
    b, c = local_logits.shape
    if total is None or total % world == 0 and b * world == (total or b * world):
        out = torch.empty((b * world, c), dtype=local_logits.dtype, device=local_logits.device)
        dist.all_gather_into_tensor(out, local_logits, group=group)
        return out
    bmax = (total + world - 1) // world
    pad = torch.zeros((bmax, c), dtype=local_logits.dtype, device=local_logits.device)
    pad[:b] = local_logits
    out = torch.empty((bmax * world, c), dtype=local_logits.dtype, device=local_logits.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    pieces = []
    for r in range(world):
        lo, hi = shard_bounds(total, r, world)
        pieces.append(out[r * bmax: r * bmax + (hi - lo)])
    return torch.cat(pieces, 0)


def sharded_predict(model, x_global):
    """ImageClassification.predict over a batch sharded across ranks: every rank returns all class ids."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    xs = shard_batch(x_global, rank, world)
    logits = model(xs)
    full = all_gather_logits(logits, total=x_global.shape[0])
    from . import tlx
    return tlx.argmax(full, axis=-1)
