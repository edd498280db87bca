"""ViT-B/16 batch 256 fp16 forward repeated under ablation bits of the fused LayerNorm + qkv kernel (tuning flavour)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tlxcv_amd
from tlxcv_amd import _lib, engine as E, models, seeded
from tlxcv_amd.models.classification import vision_transformer as V
dev = torch.device("cuda:0")
tlxcv_amd.set_precision("fp16")
m = models.vit_base_patch16_224()
m.load_dict(seeded.fill(seeded.shapes_of(m), 1))
m = m.to(dev).set_eval()
x = torch.from_numpy(seeded.image_batch(32, 0)).to(dev).repeat(8, 1, 1, 1).contiguous()


def run(self, t):          # norm1 folded, norm2 stand-alone: the variant that showed it most often
    self.attn.run(t, res=t, norm=self.norm1)
    self.mlp.run(self.norm2(t), res=t)
    return t


V.Block.run_inplace = run
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for bits in (None, 0, 0x100, 0x200, 0x400, 0x800, 0x100 | 0x400):
    def fwd():
        if bits is None:
            return m(x).clone()
        with _lib.tuning(TLXMI_DEBUG=str(bits)):
            return m(x).clone()
    ys = [fwd() for _ in range(N)]
    torch.cuda.synchronize()
    nd = sum(1 for y in ys[1:] if not torch.equal(y, ys[0]))
    print("product" if bits is None else f"tune debug=0x{bits:x}", f": {nd} of {N - 1} forwards differ from the first", flush=True)
