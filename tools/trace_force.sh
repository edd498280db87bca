export TLXMI_FORCE="25216:768:2304:1:1=11,25216:768:768:1:1=11,25216:768:3072:1:1=11,25216:3072:768:1:1=11"
export TLXMI_TRACE_TILES=1
python - <<'PY' 2>&1 | grep "^tile" | sort | uniq -c | sort -rn | head -20
import os, sys
sys.path.insert(0, os.getcwd())
import torch, tlxcv_amd
from tlxcv_amd import seeded, models, _lib
_lib.tuning().__enter__()
tlxcv_amd.set_precision("fp16")
m = models.vit_base_patch16_224(); m.load_dict(seeded.fill(seeded.shapes_of(m), 1)); m = m.to("cuda").set_eval()
x = torch.from_numpy(seeded.image_batch(32, 0)).cuda().repeat(8, 1, 1, 1).contiguous()
m(x); torch.cuda.synchronize()
PY
