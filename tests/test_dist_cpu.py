"""N>1 path on CPU: two gloo ranks exercise the shard/all-gather logic of tlxcv_amd.dist (the
exchange step of SURVEY §8e).  No engine compute here — logits are synthetic — because the product
has no CPU compute path; what is under test is ordering, ragged shards and shapes."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, q):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from tlxcv_amd import dist as D
    D.init(backend="gloo")
    full = torch.arange(total * 5, dtype=torch.float32).reshape(total, 5)       # "logits" of the whole batch
    mine = D.shard_batch(full)
    lo, hi = D.shard_bounds(total, rank, world)
    assert mine.shape[0] == hi - lo
    out = D.all_gather_logits(mine.clone(), total=total)
    q.put((rank, bool(torch.equal(out, full)), tuple(out.shape)))
    dist.barrier()
    dist.destroy_process_group()


def _worker_fwd(rank, world, port, total, q):
    """sharded_forward with a stand-in 'model' (a row-wise function, so shard results are position-independent):
    global-batch form and per-rank-shard form must give the same rows in batch order on every rank."""
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from tlxcv_amd import dist as D
    D.init(backend="gloo")
    x = torch.arange(total * 6, dtype=torch.float32).reshape(total, 6)
    model = lambda t: t[:, :4] * 2 + 1          # noqa: E731
    want = model(x)
    a = D.sharded_forward(model, x)
    lo, hi = D.shard_bounds(total, rank, world)
    b = D.sharded_forward(model, x[lo:hi], total=total)
    bad_shape = False
    try:
        D.sharded_forward(model, x[lo:hi + 1] if hi < total else x[lo:hi - 1], total=total)
    except ValueError:
        bad_shape = True
    q.put((rank, bool(torch.equal(a, want)), bool(torch.equal(b, want)), bad_shape))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [6, 5])
def test_two_rank_sharded_forward_global_and_per_rank_forms(total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_fwd, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(a and b and bad for _, a, b, bad in res), res


@pytest.mark.parametrize("total", [8, 7])
def test_two_rank_all_gather_of_logits(total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(ok for _, ok, _ in res)
    assert all(shape == (total, 5) for _, _, shape in res)


def test_shard_bounds_cover_exactly():
    from tlxcv_amd.dist import shard_bounds
    for n in (0, 1, 7, 256, 2048):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def _worker_pipe(rank, world, port, q):
    """dist.GatherPipe: step i's gather is returned by put() of step i + 1 (flush() gives the last); the staged copy
    protects a source buffer that the next step overwrites in place (a replayed hipGraph's static output)."""
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from tlxcv_amd import dist as D
    D.init(backend="gloo")
    pipe = D.GatherPipe()
    static = torch.empty((3, 4), dtype=torch.float32)          # rewritten in place every step
    got = []
    steps = 5
    for i in range(steps):
        static.copy_(torch.full((3, 4), float(100 * i + rank)))
        g = pipe.put(static)
        assert (g is None) == (i == 0)
        if g is not None:
            got.append(g.clone())
    got.append(pipe.flush().clone())
    assert pipe.flush() is None
    ok = len(got) == steps
    for i, g in enumerate(got):
        want = torch.cat([torch.full((3, 4), float(100 * i + r)) for r in range(world)], 0)
        ok = ok and bool(torch.equal(g, want))
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_pipeline_overlaps_steps_without_mixing_them():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_pipe, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(r for r, _ in res) == [0, 1] and all(ok for _, ok in res), res


def test_gather_pipe_without_a_process_group_is_a_one_step_delay():
    """Single rank: put() still returns a STAGED copy — a replayed hipGraph rewrites its static output in place before the
    previous step's result is read (the old branch returned the caller's own, already overwritten, tensor)."""
    from tlxcv_amd import dist as D
    pipe = D.GatherPipe()
    static = torch.empty(2, 3)
    static.fill_(1.0)
    assert pipe.put(static) is None
    static.fill_(2.0)                       # step 1 overwrites the static output ...
    g0 = pipe.put(static)
    assert g0 is not static and torch.equal(g0, torch.ones(2, 3))      # ... and step 0's result is still step 0's
    static.fill_(3.0)
    g1 = pipe.flush()
    assert torch.equal(g1, torch.full((2, 3), 2.0)) and pipe.flush() is None


def _worker_empty(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from tlxcv_amd import dist as D
    D.init(backend="gloo")
    raised = []
    for kw in (dict(total=None), dict(total=1)):
        x = torch.zeros((1, 4)) if kw["total"] is None else torch.zeros((1 if rank == 0 else 0, 4))
        try:
            D.sharded_forward(lambda t: t, x, **kw)
            raised.append(False)
        except ValueError:
            raised.append(True)
    q.put((rank, raised))
    dist.barrier()
    dist.destroy_process_group()


def test_a_batch_smaller_than_the_world_raises_on_every_rank_not_only_the_empty_ones():
    """1 image over 2 ranks: BOTH ranks raise before any collective (only the empty rank raised and the other one
    blocked in the all-gather until the backend's timeout)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_empty, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(r for r, _ in res) == [0, 1] and all(all(v) for _, v in res), res
