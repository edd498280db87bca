#!/usr/bin/env python3
"""bench.py — forward images/sec of the BASELINE.json workload on N MI355X GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload resnet50|vit_b16]

Workload (config.workload): BASELINE.json configs[1] — ResNet-50 fp16 forward, batch 256 per GPU,
synthetic ImageNet-shaped input already resident in HBM, seeded random weights of the real
architecture.  A "step" is one forward of one batch on every rank followed, for N>1, by the RCCL
all-gather of the (256, 1000) logits (the only exchange); scaling is weak.
Prints ONE JSON line (rank 0).  Extra objects on the same line:
  roofline      the implicit-GEMM kernel family (every conv + the classifier GEMM): algorithmic
                bytes per launch / mean launch duration, timed with HIP events on the launch stream
                in an instrumented pass right after the timed region; bound = HBM.
  cpu_baseline  the CPU oracle (oracle/functional.py, torch fp32 on all host cores) on a bounded
                sample — the stand-in for "TensorLayerX torch-CPU backend", which cannot be installed.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
MFMA_F16_PEAK_TF = 2500.0  # dense fp16 MFMA
# measured on a box of this pool (tools/roofline_denominators.py -> profiles/r01/roofline_denominators.txt):
MEASURED_HBM_COPY_GBS = 4750.0    # torch copy of 1-4 GiB, read + write bytes
MEASURED_GEMM_F16_TF = 1333.0     # hipBLASLt (torch.matmul) 16384 x 4096 x 4096


def build_model(workload, dev):
    from tlxcv_amd import models, seeded
    ctor = {"resnet50": "resnet50", "vit_b16": "vit_base_patch16_224",
            "swin_b": "swintransformer_base_patch4_window7_224"}[workload]
    m = getattr(models, ctor)()
    params = seeded.fill(seeded.shapes_of(m), 1)
    m.load_dict(params)
    return m.to(dev).set_eval(), params


def cpu_baseline(workload, params, batch=16, warm=1, min_seconds=12.0, max_iters=200):
    """Oracle restatement timed on the host cores (rank 0, N=1 only): forwards of `batch` images of the same
    workload until about `min_seconds` of CPU work have been timed (a bounded sample, 10-30 s)."""
    from oracle import functional as OF
    from tlxcv_amd import seeded
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = min(cores, 16)     # the GPU box gives one GPU's share of the host: 16 cores
    torch.set_num_threads(cores)
    p = {k: torch.from_numpy(v) for k, v in params.items()}
    x = torch.from_numpy(seeded.image_batch(batch, 0))
    fn = {"resnet50": lambda: OF.resnet(p, x, 50), "vit_b16": lambda: OF.vit(p, x, "vit_base_patch16_224"),
          "swin_b": lambda: OF.swin(p, x, "swintransformer_base_patch4_window7_224")}[workload]
    with torch.no_grad():
        for _ in range(warm):
            fn()
        t0 = time.perf_counter()
        iters = 0
        while iters < max_iters:
            fn()
            iters += 1
            dt = time.perf_counter() - t0
            if dt >= min_seconds:
                break
    return {"value": round(batch * iters / dt, 2), "unit": "images/sec", "cores": torch.get_num_threads(),
            "kind": "port", "sample": f"{iters} forwards of batch {batch}, fp32, oracle/functional.py on torch-CPU "
            f"({dt:.1f} s)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None, help="images per GPU (default 256; 128 for swin_b)")
    ap.add_argument("--workload", default="resnet50", choices=["resnet50", "vit_b16", "swin_b"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch kernel by kernel instead of replaying a hipGraph")
    a = ap.parse_args()
    if a.batch is None:
        a.batch = 128 if a.workload == "swin_b" else 256

    import tlxcv_amd
    from tlxcv_amd import dist as D, engine as E, seeded
    import torch.distributed as dist

    # rank / world from the launcher's environment; the process group (RCCL) is created only AFTER the model is built
    # and its forward captured into a hipGraph, so no communicator thread exists while the capture is open
    rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run --nproc-per-node N")
    assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU path"
    local = local % torch.cuda.device_count()   # (a 2-rank rehearsal on a 1-GPU box shares the device)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    tlxcv_amd.set_precision("fp16")

    model, params = build_model(a.workload, dev)
    x = torch.from_numpy(seeded.image_batch(min(a.batch, 32), rank)).to(dev)
    x = x.repeat((a.batch + x.shape[0] - 1) // x.shape[0], 1, 1, 1)[: a.batch].contiguous()   # resident in HBM

    fwd = model
    if not a.no_graph:
        from tlxcv_amd.graph import GraphedForward
        fwd = GraphedForward(model, x)       # the whole forward as one hipGraph; x is its static input

    if world > 1:
        r2, w2, _ = D.init()
        assert (r2, w2) == (rank, world)

    def step():
        y = fwd(x) if fwd is model else fwd()
        return D.all_gather_logits(y) if world > 1 else y

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        y = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        y = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert torch.isfinite(y.float()).all(), "non-finite logits"
    total_images = a.batch * world * a.steps
    value = total_images / dt

    line = {
        "metric": "images/sec fwd", "value": round(value, 1), "unit": "images/sec", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": {"resnet50": f"ResNet-50 fp16 forward, 224x224, batch {a.batch} per GPU (BASELINE configs[1])",
                                "vit_b16": f"ViT-B/16 fp16 forward, 224x224, batch {a.batch} per GPU (BASELINE configs[2])",
                                "swin_b": f"Swin-B (window 7) fp16 forward, 224x224, batch {a.batch} per GPU (BASELINE configs[3])"}[a.workload],
                   "global_batch": a.batch * world, "per_gpu_batch": a.batch, "weights": "seeded random (tlxcv_amd.seeded, seed 1)",
                   "parallelism": f"batch-sharded x{world}, all-gather logits" if world > 1 else "single GPU",
                   "launch": "per-kernel" if a.no_graph else "hipGraph replay of the forward"},
    }

    if rank == 0:
        # ---- roofline of the implicit-GEMM kernel family: instrumented pass, HIP events per launch
        probe = []
        E.set_probe(probe)
        nprobe = min(a.steps, 5)
        for _ in range(nprobe):
            model(x)
        torch.cuda.synchronize()
        E.set_probe(None)
        ms = sum(e[0].elapsed_time(e[1]) for e in probe)
        nl = len(probe)
        alg_bytes = sum(e[2] for e in probe)
        flops = sum(e[3] for e in probe)
        per_launch_us = 1e3 * ms / nl
        gbs = alg_bytes / (ms * 1e-3) / 1e9
        tfs = flops / (ms * 1e-3) / 1e12
        # ResNet-50's layer-by-layer arithmetic intensity (~144 FLOP/B) is below the machine balance
        # (~312): HBM-bound.  ViT-B/16 (~436 FLOP/B) is MFMA-bound.  Both fractions are reported.
        hbm_bound = a.workload == "resnet50"
        # HBM bytes per launch from PMC counters cannot be collected inside this process: they come from the
        # separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of this same command
        # (tools/pmc_traffic.sh; FETCH_SIZE doubled per MI355X_MICROARCH.md), committed under profiles/.
        traffic = None
        tj = os.path.join(REPO, "profiles", "r01", f"traffic_{a.workload}.json")
        if os.path.exists(tj) and a.batch == (128 if a.workload == "swin_b" else 256):
            tjd = json.load(open(tj))      # bytes per forward over the kernel family -> per conv2d / linear launch
            traffic = int(tjd["hbm_bytes_per_forward"] / (nl // nprobe)) if "hbm_bytes_per_forward" in tjd else int(tjd["hbm_bytes_per_launch"])
        line["roofline"] = {
            "bound": "hbm" if hbm_bound else "mfma",
            "achieved": round(gbs if hbm_bound else tfs, 1),
            "peak": HBM_PEAK_GBS if hbm_bound else MFMA_F16_PEAK_TF,
            "unit": "GB/s" if hbm_bound else "TFLOP/s",
            "frac": round(gbs / HBM_PEAK_GBS if hbm_bound else tfs / MFMA_F16_PEAK_TF, 4), "traffic": traffic,
            "peak_measured": MEASURED_HBM_COPY_GBS if hbm_bound else MEASURED_GEMM_F16_TF,
            "frac_of_measured": round(gbs / MEASURED_HBM_COPY_GBS if hbm_bound else tfs / MEASURED_GEMM_F16_TF, 4),
            "kernel": "tlxmi_conv2d kernel family: conv_igemm_kernel, gemm_pp_kernel, gemm_stream_kernel, gemm256_kernel (all instantiations; every conv / linear launch of one forward)",
            "launches_per_step": nl // nprobe, "avg_launch_us": round(per_launch_us, 2),
            "alg_bytes_per_launch": int(alg_bytes / nl), "alg_flops_per_launch": int(flops / nl),
            "hbm_gbs": round(gbs, 1), "hbm_frac": round(gbs / HBM_PEAK_GBS, 4),
            "mfma_tflops": round(tfs, 1), "mfma_frac": round(tfs / MFMA_F16_PEAK_TF, 4),
            "gemm_ms_per_step": round(ms / nprobe, 3),
        }
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(a.workload, params)
    if world > 1:
        dist.barrier()
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
