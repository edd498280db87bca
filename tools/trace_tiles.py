import os, sys
sys.path.insert(0, "/root/repo")
sys.path.insert(0, os.getcwd())
import torch, tlxcv_amd
from tlxcv_amd import seeded, models, engine as E, _lib
_lib.tuning().__enter__()
dev = torch.device("cuda:0")
tlxcv_amd.set_precision("fp16")
E.set_option("two_streams", False)
m = models.resnet50(); m.load_dict(seeded.fill(seeded.shapes_of(m), 1)); m = m.to(dev).set_eval()
x = torch.from_numpy(seeded.image_batch(32, 0)).to(dev).repeat(4, 1, 1, 1).contiguous()
m(x); torch.cuda.synchronize()
for pc in sys.argv[1:]:
    os.environ["TLXMI_PLAN_CUS"] = pc
    os.environ["TLXMI_TRACE_TILES"] = "1"
    sys.stderr.write(f"== plan {pc}\n"); sys.stderr.flush()
    m(x); torch.cuda.synchronize()
    os.environ["TLXMI_TRACE_TILES"] = "0"
