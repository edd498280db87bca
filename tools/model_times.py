"""Forward time of every model family at a realistic batch (fp16, per-kernel launches): a sanity sweep after
dispatcher changes.  usage: python tools/model_times.py [halo=1|0]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import tlxcv_amd  # noqa: E402,F401
from tlxcv_amd import seeded, models  # noqa: E402

dev = torch.device("cuda:0")
CASES = [("resnet18", 256, 224, False), ("resnet34", 256, 224, False), ("resnet50", 256, 224, False), ("resnet101", 128, 224, False),
         ("vgg16", 64, 224, False), ("alexnet", 256, 224, False), ("resnext50_32x4d", 256, 224, False),
         ("resnext101_64x4d", 64, 224, False), ("efficientnet_b0", 256, 224, False), ("resnest50", 128, 224, False),
         ("MobileNetV1", 256, 224, False), ("mobilenet_v2", 256, 224, False), ("mobilenet_v3_small", 256, 224, False),
         ("mobilenet_v3_large", 256, 224, False), ("DarkNet", 64, 256, True), ("YOLOv3", 32, 416, True),
         ("vit_small_patch16_224", 256, 224, False), ("vit_base_patch16_224", 256, 224, False),
         ("swintransformer_tiny_patch4_window7_224", 128, 224, False), ("swintransformer_base_patch4_window7_224", 128, 224, False)]
only = sys.argv[1].split(",") if len(sys.argv) > 1 else None
for ctor, bs, hw, dict_in in CASES:
    if only and ctor not in only:
        continue
    m = models.efficientnet(ctor) if ctor.startswith("efficientnet_") else getattr(models, ctor)()
    m.load_dict(seeded.fill(seeded.shapes_of(m), 1))
    m = m.to(dev).set_eval()
    x = torch.from_numpy(seeded.image_batch(min(bs, 16), 0, hw=hw)).to(dev).repeat(bs // min(bs, 16), 1, 1, 1).contiguous()
    inp = {"images": x} if dict_in else x
    for _ in range(2):
        m(inp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        m(inp)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / n
    print(f"{ctor:44s} bs{bs:4d} {hw}px  {ms:8.2f} ms  {bs / ms * 1e3:9.0f} img/s", flush=True)
    if os.environ.get("LAYERS"):      # per conv / linear launch: shape, us, GB/s, TFLOP/s (the bench's probe)
        from tlxcv_amd import engine as E
        probe = []
        E.set_probe(probe)
        m(inp)
        torch.cuda.synchronize()
        E.set_probe(None)
        agg = {}
        for e0, e1, b, f, shape in probe:
            a = agg.setdefault(shape, [0, 0.0, b, f])
            a[0] += 1
            a[1] += e0.elapsed_time(e1) * 1e3
        for shape, (cnt, us, b, f) in agg.items():
            print(f"    {str(shape):52s} x{cnt:3d} {us / cnt:8.1f} us {b / (us / cnt) / 1e3:7.0f} GB/s {f / (us / cnt) / 1e6:7.1f} TF/s")
    del m
    torch.cuda.empty_cache()
