import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
import tlxcv_amd
from tlxcv_amd import seeded, models, engine as E
dev = torch.device("cuda:0")
tlxcv_amd.set_precision("fp16")
m = models.vit_base_patch16_224()
m.load_dict(seeded.fill(seeded.shapes_of(m), 1))
m = m.to(dev).set_eval()
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
x = torch.from_numpy(seeded.image_batch(32, 0)).to(dev).repeat(bs // 32, 1, 1, 1).contiguous()
for ts in (1, 0):
    for fold in (1, 0):
        E.set_option("two_streams", ts)
        E.set_option("lnfold", fold)
        for _ in range(3):
            m(x)
        torch.cuda.synchronize()
        b0 = E._cache_builds
        t0 = time.time()
        for _ in range(5):
            m(x)
        torch.cuda.synchronize()
        print(f"two_streams={ts} lnfold={fold}: {(time.time() - t0) / 5 * 1000:.2f} ms per eager forward; cache builds during the 5: {E._cache_builds - b0}", flush=True)
