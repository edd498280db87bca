#!/usr/bin/env python3
"""bench.py — forward images/sec of the BASELINE.json workloads on N MI355X GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload resnet50|vit_b16|swin_b] [--no-also]

Headline (config.workload): BASELINE.json configs[1] — ResNet-50 fp16 forward, batch 256 per GPU, synthetic
ImageNet-shaped input already resident in HBM, seeded random weights of the real architecture.  A "step" is one forward
of one batch on every rank followed, for N>1, by the RCCL all-gather of the (256, 1000) logits (the only exchange,
SURVEY.md §8e); scaling is weak.  Prints ONE JSON line (rank 0).  On the same line:

  also          (N=1, default workload) the other half of the metric and the third config, each measured the same way in
                this same run: ViT-B/16 batch 256 (configs[2]) and Swin-B batch 128 (configs[3]), each with
                value / ms_per_step / roofline.
  roofline      the WHOLE FORWARD over the TIMED region: SURVEY.md §8d's algorithmic bytes (or FLOPs) per image x the
                images of a step (+ the weights once per step), divided by the timed ms_per_step — so achieved and
                value come from the same clock.  `kernel_family` (secondary) is the implicit-GEMM kernel family (every
                conv / linear launch): Σ algorithmic bytes and FLOPs per launch from the launch descriptors, durations
                from HIP events on the launch stream in an instrumented per-kernel pass; its share of that pass's
                wall time is applied to the timed step, so family_ms_per_step <= ms_per_step by construction.
                `traffic`: HBM bytes per forward from separate rocprofv3 --pmc passes (tools/pmc_traffic.sh), only when
                the file under profiles/ was measured on exactly these kernel sources (sha of csrc/), else null.
  cpu_baseline  the CPU oracle (oracle/functional.py, torch fp32 on the host cores) on a bounded sample — the stand-in
                for "TensorLayerX torch-CPU backend", which cannot be installed.  ResNet-50 at top level, ViT-B/16 inside
                also[0] (both halves of the headline metric); threads = min(cores the process may use, 16): a GPU box
                hands one GPU's share of the host, 16 cores, whatever os.cpu_count() says.  `all_affinity_cores` (ResNet-50
                only): the same protocol on every core of the affinity mask, in a CPU-only child process with a 75 s budget
                (value null + the reason when the box cannot finish it: see cpu_baseline()).
Progress lines go to stderr ([bench] ...), the JSON line alone to stdout; the default run takes 2 - 4 minutes.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402,F401
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
HBM_ACHIEVABLE_GBS = 6290.0      # same guide: 6.29 TB/s measured float4 copy (79 % of spec)
MFMA_F16_PEAK_TF = 2500.0        # dense fp16 MFMA, spec
MEASURED_GEMM_F16_TF = 1333.0    # hipBLASLt (torch.matmul) 16384 x 4096 x 4096 on a box of this pool (profiles/r01/roofline_denominators.txt)
PROFILE_ROUND = "r05"

# SURVEY.md §8d / BASELINE.md §2: algorithmic work per image with every elementwise op fused into its producer
WORK = {
    #            ctor                                        batch  FLOP/img   bytes/img  weight bytes  bound
    "resnet50": ("resnet50", 256, 8.178e9, 56.8e6, 51.0e6, "hbm"),
    "vit_b16": ("vit_base_patch16_224", 256, 35.128e9, 79.9e6, 173e6, "mfma"),
    "swin_b": ("swintransformer_base_patch4_window7_224", 128, 30.862e9, 137e6, 176e6, "hbm"),
}
LABEL = {"resnet50": "ResNet-50 fp16 forward, 224x224, batch {b} per GPU (BASELINE configs[1])",
         "vit_b16": "ViT-B/16 fp16 forward, 224x224, batch {b} per GPU (BASELINE configs[2])",
         "swin_b": "Swin-B (window 7) fp16 forward, 224x224, batch {b} per GPU (BASELINE configs[3])"}


def csrc_sha():
    """sha of the kernel sources: a PMC traffic file is only quoted for the build it was measured on."""
    h = hashlib.sha256()
    d = os.path.join(REPO, "tlxcv_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def build_model(workload, dev):
    from tlxcv_amd import models, seeded
    m = getattr(models, WORK[workload][0])()
    params = seeded.fill(seeded.shapes_of(m), 1)
    m.load_dict(params)
    return m.to(dev).set_eval(), params


def _cpu_forward_fn(workload, params, batch):
    from oracle import functional as OF
    from tlxcv_amd import seeded
    p = {k: torch.from_numpy(v) for k, v in params.items()}
    x = torch.from_numpy(seeded.image_batch(batch, 0))
    return {"resnet50": lambda: OF.resnet(p, x, 50), "vit_b16": lambda: OF.vit(p, x, "vit_base_patch16_224"),
            "swin_b": lambda: OF.swin(p, x, "swintransformer_base_patch4_window7_224")}[workload]


def _cpu_timed(fn, threads, batch, warm, min_iters, min_seconds, max_iters=200):
    torch.set_num_threads(threads)
    with torch.no_grad():
        for _ in range(warm):
            fn()
        t0 = time.perf_counter()
        iters = 0
        while iters < max_iters:
            fn()
            iters += 1
            dt = time.perf_counter() - t0
            if iters >= min_iters and dt >= min_seconds:
                break
    return round(batch * iters / dt, 2), torch.get_num_threads(), iters, dt


def _affinity():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def cpu_baseline(workload, params, batch=32, warm=2, min_iters=5, min_seconds=8.0, all_cores=True, all_cores_budget_s=75):
    """Oracle restatement timed on the host cores (rank 0, N=1 only): BASELINE.md §4 — batch 32, fp32, 2 warm-up forwards, then at least
    5 timed forwards and at least ~8 s of CPU work (a bounded sample).  `value` / `cores`: min(cores this process may run on, 16)
    threads — a GPU box grants one GPU's share of its host, 16 CPUs' worth of time, whatever the affinity mask shows.
    `all_affinity_cores` (north_star "all host cores, count stated"): the same protocol on EVERY core of the affinity
    mask, in a CPU-only child process with a wall-clock budget — on a box whose mask shows 256 cores and whose cgroup grants 16, 256
    threads spend their time being descheduled and a single forward can take minutes; the child is then stopped and the field says so
    instead of holding up the bench line."""
    avail = _affinity()
    fn = _cpu_forward_fn(workload, params, batch)
    print(f"[bench] cpu_baseline {workload}: {min(avail, 16)} threads ...", file=sys.stderr, flush=True)
    v, c, it, dt = _cpu_timed(fn, min(avail, 16), batch, warm, min_iters, min_seconds)
    out = {"value": v, "unit": "images/sec", "cores": c, "kind": "port",
           "sample": f"{it} forwards of batch {batch} after {warm} warm-up, fp32, oracle/functional.py on torch-CPU ({dt:.1f} s), "
                     f"{c} threads (the process may run on {avail} cores; os.cpu_count() = {os.cpu_count()})"}
    if all_cores and avail > 16:
        print(f"[bench] cpu_baseline {workload}: {avail} threads in a child process (budget {all_cores_budget_s} s) ...", file=sys.stderr, flush=True)
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", workload, str(avail)]
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=all_cores_budget_s)
            got = json.loads(r.stdout.strip().splitlines()[-1])
            out["all_affinity_cores"] = got
        except subprocess.TimeoutExpired:
            out["all_affinity_cores"] = {"value": None, "cores": avail,
                                         "sample": f"2 warm-up + 2 timed forwards of batch {batch} on {avail} threads did not finish within {all_cores_budget_s} s "
                                                   f"(the 16-thread run above needs {dt / it:.1f} s a forward): the affinity mask is wider than the CPU time this box grants"}
        except Exception as e:      # noqa: BLE001 — the bench line must not die on the secondary figure
            out["all_affinity_cores"] = {"value": None, "cores": avail, "sample": f"child failed after {time.perf_counter() - t0:.0f} s: {type(e).__name__}: {e}"[:300]}
    return out


def cpu_baseline_child(workload, threads, batch=32):
    """`bench.py --cpu-baseline-child WORKLOAD THREADS`: CPU only (never touches the GPU), prints one JSON object."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from tlxcv_amd import models, seeded
    m = getattr(models, WORK[workload][0])()
    params = seeded.fill(seeded.shapes_of(m), 1)
    del m
    fn = _cpu_forward_fn(workload, params, batch)
    v, c, it, dt = _cpu_timed(fn, threads, batch, 2, 2, 5.0)
    print(json.dumps({"value": v, "cores": c, "sample": f"{it} forwards of batch {batch} after 2 warm-up ({dt:.1f} s) on {c} threads, child process"}), flush=True)


def measure(workload, batch, steps, warmup, dev, rank, world, graph=True, probe_family=True):
    """Timed region + roofline of one workload on this rank.  Returns (fields, params, model)."""
    from tlxcv_amd import dist as D, engine as E, seeded
    import torch.distributed as dist
    model, params = build_model(workload, dev)
    x = torch.from_numpy(seeded.image_batch(min(batch, 32), rank)).to(dev)
    x = x.repeat((batch + x.shape[0] - 1) // x.shape[0], 1, 1, 1)[:batch].half().contiguous()   # resident in HBM, fp16 as BASELINE.md 3 names it ("seed 0, cast fp16")

    fwd = model
    if graph:
        from tlxcv_amd.graph import GraphedForward
        fwd = GraphedForward(model, x)       # the whole forward as one hipGraph; x is its static input

    if world > 1 and not dist.is_initialized():
        # the process group (RCCL) is created only AFTER the forward is captured, so no communicator thread exists
        # while the capture is open
        r2, w2, _ = D.init()
        assert (r2, w2) == (rank, world)

    # N > 1: the all-gather of step i runs on RCCL's stream under the forward of step i + 1 (dist.GatherPipe); every step's
    # gather is complete inside the timed region (fence() flushes the last one before the barrier)
    pipe = D.GatherPipe() if world > 1 else None
    last = [None]
    pipe_ok = [True]

    def step():
        y = fwd(x) if fwd is model else fwd()
        if pipe is None:
            return y
        if pipe_ok[0]:
            try:
                g = pipe.put(y)
            except Exception as e:                      # an RCCL build without async all_gather_into_tensor: gather in step
                print(f"[bench] GatherPipe unavailable ({e!r}); synchronous all-gather per step", file=sys.stderr, flush=True)
                pipe_ok[0] = False
                g = D.all_gather_logits(y)
        else:
            g = D.all_gather_logits(y)
        if g is not None:
            last[0] = g
        return y

    def fence():
        if world > 1:
            g = pipe.flush()
            if g is not None:
                last[0] = g
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        y = step()
    fence()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(steps):
        y = step()
        marks[i + 1].record()               # one event record per step on the launch stream (no host sync)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert torch.isfinite(y.float()).all(), "non-finite logits"
    if world > 1:
        assert last[0] is not None and last[0].shape[0] == batch * world and torch.isfinite(last[0].float()).all(), "gathered logits"
        assert torch.equal(last[0][rank * batch:(rank + 1) * batch].to(y.device), y), "this rank's rows of the gathered logits"
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
    median_ms = per_step[steps // 2] if steps % 2 else 0.5 * (per_step[steps // 2 - 1] + per_step[steps // 2])
    ms_per_step = 1e3 * dt / steps
    value = batch * world * steps / dt
    out = {"value": round(value, 1), "ms_per_step": round(ms_per_step, 4),
           "ms_per_step_median": round(median_ms, 4), "value_at_median": round(batch * world / (median_ms * 1e-3), 1)}

    if rank == 0:
        _, _, flop_img, bytes_img, wbytes, bound = WORK[workload]
        scale = batch                                    # per GPU: rank 0's own step
        alg_bytes = bytes_img * scale + wbytes
        alg_flops = flop_img * scale
        gbs = alg_bytes / (ms_per_step * 1e-3) / 1e9
        tfs = alg_flops / (ms_per_step * 1e-3) / 1e12
        hbm = bound == "hbm"
        roof = {
            "bound": bound, "achieved": round(gbs if hbm else tfs, 1), "peak": HBM_PEAK_GBS if hbm else MFMA_F16_PEAK_TF,
            "unit": "GB/s" if hbm else "TFLOP/s", "frac": round(gbs / HBM_PEAK_GBS if hbm else tfs / MFMA_F16_PEAK_TF, 4),
            "traffic": None,
            "scope": "whole forward of one step over the timed region (algorithmic work of SURVEY.md 8d / timed ms_per_step)",
            "alg_bytes_per_step": int(alg_bytes), "alg_flops_per_step": int(alg_flops),
            "hbm_gbs": round(gbs, 1), "hbm_frac": round(gbs / HBM_PEAK_GBS, 4),
            "hbm_frac_of_achievable_6290": round(gbs / HBM_ACHIEVABLE_GBS, 4),
            "mfma_tflops": round(tfs, 1), "mfma_frac": round(tfs / MFMA_F16_PEAK_TF, 4),
            "mfma_frac_of_hipblaslt_1333": round(tfs / MEASURED_GEMM_F16_TF, 4),
            "frac_at_median_step": round((gbs / HBM_PEAK_GBS if hbm else tfs / MFMA_F16_PEAK_TF) * ms_per_step / median_ms, 4),
        }
        tj = os.path.join(REPO, "profiles", PROFILE_ROUND, f"traffic_{workload}.json")
        if os.path.exists(tj) and batch == WORK[workload][1]:
            tjd = json.load(open(tj))
            if tjd.get("csrc_sha") == csrc_sha():        # measured on exactly these kernels, else stale -> null
                roof["traffic"] = int(tjd["hbm_bytes_per_forward"])
                roof["traffic_source"] = f"profiles/{PROFILE_ROUND}/traffic_{workload}.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes, all kernels of a forward)"
                roof["traffic_over_algorithmic"] = round(roof["traffic"] / alg_bytes, 3)
        if probe_family:
            # ---- secondary: the implicit-GEMM kernel family, HIP event pair per launch in a per-kernel pass
            probe = []
            nprobe = 3
            model(x)
            torch.cuda.synchronize()
            w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            E.set_probe(probe)
            w0.record()
            for _ in range(nprobe):
                model(x)
            w1.record()
            torch.cuda.synchronize()
            E.set_probe(None)
            fam_ms = sum(e[0].elapsed_time(e[1]) for e in probe)
            wall_ms = w0.elapsed_time(w1)
            nl = len(probe)
            share = min(1.0, fam_ms / wall_ms)
            fam_step_ms = share * ms_per_step
            fb = sum(e[2] for e in probe) / nprobe
            ff = sum(e[3] for e in probe) / nprobe
            roof["kernel_family"] = {
                "kernel": "tlxmi_conv2d family: conv_igemm / conv_halo / gemm_pp / gemm_stream / gemm256 / fused-block kernels (every conv / linear launch of one forward)",
                "launches_per_step": nl // nprobe, "share_of_forward_time": round(share, 4),
                "family_ms_per_step": round(fam_step_ms, 3), "avg_launch_us": round(1e3 * fam_step_ms / (nl // nprobe), 2),
                "alg_bytes_per_launch": int(fb / (nl // nprobe)), "alg_flops_per_launch": int(ff / (nl // nprobe)),
                "hbm_gbs": round(fb / (fam_step_ms * 1e-3) / 1e9, 1), "mfma_tflops": round(ff / (fam_step_ms * 1e-3) / 1e12, 1),
                "frac": round((fb / (fam_step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if hbm else (ff / (fam_step_ms * 1e-3) / 1e12 / MFMA_F16_PEAK_TF), 4),
                "method": "per-launch HIP event pairs in an instrumented per-kernel pass; the family's share of that pass's wall time x the timed ms_per_step",
            }
        out["roofline"] = roof
        if graph and world == 1 and probe_family:
            # ---- secondary: consecutive steps on two alternating launch streams (two batches in flight: the ramp-down of one
            # forward overlaps the ramp-up of the next).  NOT the reported value: `value` stays one step at a time.
            from tlxcv_amd.graph import GraphedForward
            g2 = GraphedForward(model, x.clone())
            ls = (torch.cuda.Stream(), torch.cuda.Stream())
            fw = (fwd, g2)
            n2 = max(10, min(steps, 30))

            def burst(n):
                for i in range(n):
                    with torch.cuda.stream(ls[i & 1]):
                        fw[i & 1]()
            cur = torch.cuda.current_stream()
            for s_ in ls:
                s_.wait_stream(cur)
            burst(4)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            burst(n2)
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t0
            out["two_in_flight"] = {"value": round(batch * n2 / dt2, 1), "ms_per_step": round(1e3 * dt2 / n2, 4), "steps": n2,
                                    "note": "same forward, consecutive steps alternate between two launch streams; secondary figure"}
    return out, params, model


def main():
    if len(sys.argv) >= 4 and sys.argv[1] == "--cpu-baseline-child":
        cpu_baseline_child(sys.argv[2], int(sys.argv[3]))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=None, help="images per GPU (default 256; 128 for swin_b)")
    ap.add_argument("--workload", default="resnet50", choices=list(WORK))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="headline workload only (skip the ViT-B/16 and Swin-B lines)")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=0|1", help="engine.set_option(NAME, value) before building")
    ap.add_argument("--no-probe", action="store_true", help="timed region only: no instrumented per-kernel pass, no two_in_flight run (profiling runs)")
    ap.add_argument("--no-graph", action="store_true", help="launch kernel by kernel instead of replaying a hipGraph")
    a = ap.parse_args()
    if a.batch is None:
        a.batch = WORK[a.workload][1]

    import tlxcv_amd
    import torch.distributed as dist

    rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run --nproc-per-node N")
    assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU path"
    local = local % torch.cuda.device_count()   # (a 2-rank rehearsal on a 1-GPU box shares the device)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    tlxcv_amd.set_precision("fp16")
    for o in a.option:
        name, _, val = o.partition("=")
        tlxcv_amd.engine.set_option(name, int(val or "1"))

    if rank == 0:
        print(f"[bench] {a.workload} batch {a.batch} x {world} GPU(s), {a.steps} steps ...", file=sys.stderr, flush=True)
    res, params, model = measure(a.workload, a.batch, a.steps, a.warmup, dev, rank, world, graph=not a.no_graph, probe_family=not a.no_probe)
    line = {
        "metric": "images/sec fwd", "value": res["value"], "unit": "images/sec", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": res["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "ms_per_step_median": res["ms_per_step_median"], "value_at_median": res["value_at_median"],
        **({"two_in_flight": res["two_in_flight"]} if "two_in_flight" in res else {}),
        "config": {"workload": LABEL[a.workload].format(b=a.batch),
                   "global_batch": a.batch * world, "per_gpu_batch": a.batch, "weights": "seeded random (tlxcv_amd.seeded, seed 1)",
                   "input": "(B, 3, 224, 224) NCHW fp16 resident in HBM (BASELINE.md 3: standard-normal, cast fp16)",
                   "parallelism": f"batch-sharded x{world}, all-gather logits" if world > 1 else "single GPU",
                   "launch": "per-kernel" if a.no_graph else "hipGraph replay of the forward",
                   "csrc_sha": csrc_sha()},
    }
    if rank == 0:
        line["roofline"] = res["roofline"]
        if world == 1 and a.workload == "resnet50" and not a.no_also:
            del model
            torch.cuda.empty_cache()
            also = []
            for wl in ("vit_b16", "swin_b"):
                b = WORK[wl][1]
                print(f"[bench] also: {wl} batch {b} ...", file=sys.stderr, flush=True)
                r, p2, m2 = measure(wl, b, min(a.steps, 30), min(a.warmup, 5), dev, 0, 1, graph=not a.no_graph)
                del m2
                torch.cuda.empty_cache()
                also.append({"workload": LABEL[wl].format(b=b), "value": r["value"], "unit": "images/sec",
                             "ms_per_step": r["ms_per_step"], "ms_per_step_median": r["ms_per_step_median"],
                             "steps": min(a.steps, 30), "warmup": min(a.warmup, 5), "roofline": r["roofline"],
                             **({"two_in_flight": r["two_in_flight"]} if "two_in_flight" in r else {})})
                if wl == "vit_b16" and not a.no_cpu_baseline:
                    # the other half of the headline metric gets its CPU number in the same run
                    also[-1]["cpu_baseline"] = cpu_baseline(wl, p2, min_iters=3, min_seconds=8.0, all_cores=False)
                del p2
            line["also"] = also
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
