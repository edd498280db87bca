"""Throughput against batch: ResNet-50 / ViT-B/16 / Swin-B at batch 1, 8, 32, 64, 96, 128, 256, 512 on the hipGraph replay, with the
dispatch switch points of each batch marked AND measured: every host-side switch that changes at some batch is toggled on the same graph
replay — two_streams (forwards of >= 96 / 32 / 32 images), the bottleneck seams (>= 12 images per launch; the 14 x 14 ones from 96 to 192 images
per launch), split-K (classifier head <= 512 rows, 7 x 7 convs with few tiles), the folded LayerNorm (>= 2048 token rows per launch).
Output -> profiles/<round>/batch_table.txt.   usage: batch_table.py [models] [batches]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import tlxcv_amd  # noqa: E402
from tlxcv_amd import seeded, models, engine as E  # noqa: E402
from tlxcv_amd.graph import GraphedForward  # noqa: E402

dev = torch.device("cuda:0")
tlxcv_amd.set_precision("fp16")
CT = {"resnet50": "resnet50", "vit_b16": "vit_base_patch16_224", "swin_b": "swintransformer_base_patch4_window7_224"}
# the switches worth an arm per model: name -> (option, off value)
ARMS = {"resnet50": {"one stream": ("two_streams", 0), "no seams": ("seams", 0), "no 14x14 seams": ("seam256", 0), "no split-K": ("splitk", 0),
                     "no conv split-K": ("conv_splitk", 0)},
        "vit_b16": {"one stream": ("two_streams", 0), "LayerNorm launches": ("lnfold", 0)},
        "swin_b": {"one stream": ("two_streams", 0), "LayerNorm / window passes": ("lnfold", 0)}}
wls = sys.argv[1].split(",") if len(sys.argv) > 1 else list(CT)
batches = [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else "1,8,32,64,96,128,256,512".split(","))]
for wl in wls:
    m = getattr(models, CT[wl])()
    m.load_dict(seeded.fill(seeded.shapes_of(m), 1))
    m = m.to(dev).set_eval()
    print(f"== {wl} (fp16, hipGraph replay; ms per forward, images/s of the product dispatch, then each switch turned OFF on the same box)", flush=True)
    for bs in batches:
        x = torch.from_numpy(seeded.image_batch(min(bs, 16), 0)).to(dev).repeat((bs + 15) // 16, 1, 1, 1)[:bs].half().contiguous()
        graphs = {"product": GraphedForward(m, x)}
        for name, (opt, off) in ARMS[wl].items():
            keep = E.option(opt)
            E.set_option(opt, off)
            try:
                graphs[name] = GraphedForward(m, x)
            finally:
                E.set_option(opt, keep)
        ts = {k: [] for k in graphs}
        for r in range(5):
            for k, f in graphs.items():
                f(); torch.cuda.synchronize()
                n = 20 if bs <= 128 else 10
                t0 = time.perf_counter()
                for _ in range(n):
                    f()
                torch.cuda.synchronize()
                ts[k].append(1e3 * (time.perf_counter() - t0) / n)
        med = {k: sorted(v)[2] for k, v in ts.items()}
        base = med["product"]
        cells = "   ".join(f"{k}: {v:.3f} ({100 * (v - base) / base:+.1f} %)" for k, v in med.items() if k != "product")
        print(f"batch {bs:4d}: {base:8.3f} ms  {bs / base * 1e3:8.0f} img/s   | {cells}", flush=True)
        del graphs
    del m
    torch.cuda.empty_cache()
