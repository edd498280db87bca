"""ViT-B/16 batch 256 fp16: are repeated forwards bit-identical?  Variants: both LayerNorms folded, only norm1, only norm2, none."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tlxcv_amd
from tlxcv_amd import engine as E, models, seeded
from tlxcv_amd.models.classification import vision_transformer as V
dev = torch.device("cuda:0")
tlxcv_amd.set_precision("fp16")
m = models.vit_base_patch16_224()
m.load_dict(seeded.fill(seeded.shapes_of(m), 1))
m = m.to(dev).set_eval()
x = torch.from_numpy(seeded.image_batch(32, 0)).to(dev).repeat(8, 1, 1, 1).contiguous()
orig = V.Block.run_inplace


def variant(n1, n2):
    def run(self, t):
        self.attn.run(t, res=t, norm=self.norm1) if n1 else self.attn.run(self.norm1(t), res=t)
        self.mlp.run(t, res=t, norm=self.norm2) if n2 else self.mlp.run(self.norm2(t), res=t)
        return t
    return run


for name, n1, n2 in (("both", 1, 1), ("norm1 only", 1, 0), ("norm2 only", 0, 1), ("none", 0, 0)):
    V.Block.run_inplace = variant(n1, n2)
    ys = [m(x).clone() for _ in range(5)]
    torch.cuda.synchronize()
    diffs = [int((y != ys[0]).sum()) for y in ys[1:]]
    print(name, "elements differing from run 0:", diffs, "max|d|", max(float((y.float() - ys[0].float()).abs().max()) for y in ys[1:]))
