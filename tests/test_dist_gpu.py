"""BASELINE.json configs[4]'s code path with real engine output: two fresh child ranks on the box's one GPU run
tlxcv_amd.dist.sharded_forward / sharded_predict on the resnet50_b4 golden images (2 + 2 images, and a ragged 2 + 1
split handed over as per-rank shards), all-gather the logits, and BOTH ranks must hold the golden logits / class ids.
The collective is gloo here (TLXMI_DIST_BACKEND=gloo: RCCL refuses two ranks on one device); on the 8-GPU node the same
code runs over RCCL (bench.py --gpus N)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, REPO
from util import check_fp16_logits

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_on_one_gpu_gather_the_golden_logits(dev, tmp_path):
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), TLXMI_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, "tests", "dist_gpu_worker.py"), str(tmp_path)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o.decode(errors="replace"))
    for rank, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} failed:\n{o[-3000:]}"
    g = np.load(os.path.join(GOLDEN, "resnet50_b4.npz"))
    ref = g["logits"]
    for rank in range(2):
        r = np.load(os.path.join(str(tmp_path), f"rank{rank}.npz"))
        assert r["logits_global4"].shape == (4, 1000) and r["logits_shard3"].shape == (3, 1000)
        assert np.abs(r["logits_global4"] - ref).max() <= 1e-4, rank             # north_star: 1e-4 fp32, on every rank
        assert np.abs(r["logits_shard3"] - ref[:3]).max() <= 1e-4, rank
        assert (r["pred_global4"] == g["argmax"]).all(), rank                    # bit-exact class indices
        check_fp16_logits(r["logits_fp16"], ref, g["argmax"], "resnet50_b4")
        # gather pipeline over a replayed graph (fp16): steps 0 and 2 = the batch in order, step 1 = each rank's shard flipped
        ps = r["pipe_steps"]
        assert ps.shape == (3, 4, 1000)
        check_fp16_logits(ps[0], ref, g["argmax"], "resnet50_b4")
        assert np.array_equal(ps[2], ps[0])
        check_fp16_logits(ps[1], ref[[1, 0, 3, 2]], g["argmax"][[1, 0, 3, 2]], "resnet50_b4")
    a, b = (np.load(os.path.join(str(tmp_path), f"rank{k}.npz")) for k in range(2))
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k                                     # the ranks agree bit for bit


def test_bench_py_two_ranks_rehearsal_prints_one_line_with_gathered_logits(dev):
    """VERDICT r2 #1d: the driver's first multi-GPU run of `bench.py --gpus N` must not be the first run of that code path.
    Two fresh child ranks (started before any GPU call of theirs) share the box's one GPU over gloo and run the command the
    driver runs: graph capture -> process group -> GatherPipe -> barrier-bracketed timing -> ONE JSON line from rank 0;
    bench.py itself asserts that the gathered logits are finite and that each rank's rows are its own logits."""
    import json
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), TLXMI_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                                       "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((o.decode(errors="replace"), e.decode(errors="replace")))
    for rank, (p, (o, e)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} failed:\n{o[-1500:]}\n{e[-3000:]}"
    lines = [ln for ln in outs[0][0].splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and not [ln for ln in outs[1][0].splitlines() if ln.startswith("{")]      # rank 0 only, one line
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "weak" and line["unit"] == "images/sec"
    assert line["config"]["global_batch"] == 512 and line["value"] > 0 and line["roofline"]["frac"] > 0


def test_one_rank_on_the_real_rccl_backend_gathers_beside_a_replaying_graph(dev, tmp_path):
    """RCCL executed once on the one GPU this box has.  A fresh child process initialises the "nccl" (= RCCL) backend at
    world size 1 and drives dist.GatherPipe(force_collective=True) — librccl loaded, a communicator created, the asynchronous
    all_gather_into_tensor on RCCL's stream next to hipGraph replays on the caller's — over six steps whose inputs alternate; every
    gathered block must be that step's logits.  (No scaling is measured here: that is the driver's 8-GPU run.)"""
    port = _free_port()
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("TLXMI_DIST_BACKEND", None)
    p = subprocess.Popen([sys.executable, os.path.join(REPO, "tests", "dist_rccl1_worker.py"), str(tmp_path)], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    try:
        o, _ = p.communicate(timeout=600)
    except subprocess.TimeoutExpired:
        p.kill()
        raise
    o = o.decode(errors="replace")
    assert p.returncode == 0 and "RCCL_ONE_RANK_OK" in o, o[-3000:]
    g = np.load(os.path.join(GOLDEN, "resnet50_b4.npz"))
    ref = g["logits"]
    r = np.load(os.path.join(str(tmp_path), "rccl1.npz"))
    check_fp16_logits(r["logits_sharded"], ref, g["argmax"], "resnet50_b4")
    ps = r["pipe_steps"]
    assert ps.shape == (6, 4, 1000)
    for i in range(6):
        want, am = (ref, g["argmax"]) if i % 2 == 0 else (ref[::-1], g["argmax"][::-1])
        check_fp16_logits(ps[i], want, am, "resnet50_b4")
    assert np.array_equal(ps[0], ps[2]) and np.array_equal(ps[1], ps[3])          # replays are bit-reproducible through the gather
    check_fp16_logits(r["gather_rows"], ref, g["argmax"], "resnet50_b4")
