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
    if (world > 1 or os.environ.get("TLXMI_DIST_INIT_SINGLE") == "1") and not dist.is_initialized():      # (single: the one-rank RCCL rehearsal)
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
        return _gather_rows(local_logits, group)
    bmax = (total + world - 1) // world
    pad = torch.zeros((bmax, c), dtype=local_logits.dtype, device=local_logits.device)
    pad[:b] = local_logits
    out = _gather_rows(pad, group)
    pieces = []
    for r in range(world):
        lo, hi = shard_bounds(total, r, world)
        pieces.append(out[r * bmax: r * bmax + (hi - lo)])
    return torch.cat(pieces, 0)


def _gather_rows(padded, group=None):
    """all_gather_into_tensor of equal (rows, classes) blocks.  RCCL gathers device tensors in place; the gloo rehearsal
    backend (several ranks sharing one GPU, or CPU tests) stages device tensors through the host."""
    world = dist.get_world_size(group)
    b, c = padded.shape
    if padded.is_cuda and dist.get_backend(group) == "gloo":
        host = padded.cpu()
        out = torch.empty((b * world, c), dtype=host.dtype)
        dist.all_gather_into_tensor(out, host, group=group)
        return out.to(padded.device)
    out = torch.empty((b * world, c), dtype=padded.dtype, device=padded.device)
    dist.all_gather_into_tensor(out, padded, group=group)
    return out


class GatherPipe:
    """The all-gather of step i overlapped with the forward of step i + 1 (equal shards).  put(y) copies this rank's logits
    into one of two staging buffers (a forward that replays a hipGraph rewrites its static output every step) and starts the
    all-gather asynchronously — RCCL runs it on its own stream, the caller's stream goes on to the next forward — then waits
    for the PREVIOUS step's gather and returns that result (None on the first call).  flush() returns the last one.
    With the gloo rehearsal backend (device tensors staged through the host) every gather is synchronous."""

    def __init__(self, group=None, force_collective=False):
        """force_collective: take the collective branch also at world size 1 (an initialised process group is required) — a one-rank
        RCCL communicator still loads librccl, creates the communicator and runs the asynchronous all_gather_into_tensor on RCCL's
        stream beside the caller's hipGraph replays: tests/test_dist_gpu.py runs the real backend that way on a one-GPU box."""
        self.group = group
        self.force = bool(force_collective)
        self.stage = [None, None]
        self.n = 0
        self.pending = None        # (out, work)

    def _wait(self):
        if self.pending is None:
            return None
        out, work = self.pending
        self.pending = None
        if work is not None:
            work.wait()
        return out

    def put(self, local_logits):
        if not dist.is_initialized() or (dist.get_world_size(self.group) == 1 and not self.force):
            # single rank: still a staged copy — the caller's next replay rewrites `local_logits` before this result is read
            k = self.n & 1
            self.n += 1
            if self.stage[k] is None or self.stage[k].shape != local_logits.shape or self.stage[k].dtype != local_logits.dtype:
                self.stage[k] = torch.empty_like(local_logits, memory_format=torch.contiguous_format)
            self.stage[k].copy_(local_logits)
            prev, self.pending = self._wait(), (self.stage[k], None)
            return prev
        world = dist.get_world_size(self.group)
        if local_logits.is_cuda and dist.get_backend(self.group) == "gloo":
            prev, self.pending = self._wait(), (_gather_rows(local_logits.contiguous(), self.group), None)
            return prev
        k = self.n & 1
        self.n += 1
        if self.stage[k] is None or self.stage[k].shape != local_logits.shape or self.stage[k].dtype != local_logits.dtype:
            self.stage[k] = torch.empty_like(local_logits, memory_format=torch.contiguous_format)
        self.stage[k].copy_(local_logits)
        b, c = local_logits.shape
        out = torch.empty((b * world, c), dtype=local_logits.dtype, device=local_logits.device)
        work = dist.all_gather_into_tensor(out, self.stage[k], group=self.group, async_op=True)
        prev = self._wait()
        self.pending = (out, work)
        return prev

    def flush(self):
        return self._wait()


def sharded_forward(model, x, total=None):
    """Logits of a batch sharded across ranks, on every rank, in batch order.
    total=None: `x` is the GLOBAL batch (every rank holds it, e.g. a broadcast request) and each rank runs its
    contiguous shard; total=n: `x` is already THIS RANK'S shard of an n-image batch (shard_bounds(n, rank, world)) —
    the form a data loader that reads per rank uses, no rank ever holds the other ranks' images."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    if total is not None and total < world:
        raise ValueError(f"batch of {total} images over {world} ranks leaves empty shards; use fewer ranks")
    if total is None:
        total = x.shape[0]
        xs = shard_batch(x, rank, world)
    else:
        lo, hi = shard_bounds(total, rank, world)
        if x.shape[0] != hi - lo:
            raise ValueError(f"rank {rank} of {world}: shard has {x.shape[0]} images, shard_bounds({total}) says {hi - lo}")
        xs = x
    if total < world:
        # every rank knows total and world: all of them raise, none is left waiting in the collective
        raise ValueError(f"batch of {total} images over {world} ranks leaves empty shards; use fewer ranks")
    return all_gather_logits(model(xs), total=total)


def sharded_predict(model, x, total=None):
    """ImageClassification.predict over a batch sharded across ranks: every rank returns all class ids
    (arguments as sharded_forward)."""
    from . import tlx
    return tlx.argmax(sharded_forward(model, x, total), axis=-1)
