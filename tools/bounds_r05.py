"""Round-5 upper bounds, each on the hipGraph replay of the WHOLE two-stream forward (tuning flavour; results of the ablated arms are wrong,
timing only), replays interleaved in one process:
  ViT-B/16 batch 256:  qkv stores dropped (TLXMI_DEBUG=2 on the launches with Cout 2304), attention K / V staging dropped (TLXMI_ATTN_DBG=1),
                       both = what a qkv + attention fusion can remove at the very most; the 24 LayerNorm launches dropped = the bound of
                       LayerNorm statistics from the producing GEMM.
  Swin-B batch 128:    the two LayerNorm-type passes of every block dropped = the bound of the whole-row GEMM epilogue.
usage: bounds_r05.py [vit|swin|all]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tlxcv_amd
from tlxcv_amd import seeded, models, engine as E, _lib
from tlxcv_amd.tlx import nn
_lib.tuning().__enter__()

dev = torch.device("cuda:0")
tlxcv_amd.set_precision("fp16")
which = sys.argv[1] if len(sys.argv) > 1 else "all"

_ln_fwd = nn.LayerNorm.forward
_lnwp, _wrln = E.layernorm_window_partition, E.window_reverse_layernorm


def skip_ln(on):
    if on:
        nn.LayerNorm.forward = lambda self, x: x
        E.layernorm_window_partition = lambda x, g, b, eps, ws, shift: x.view(x.shape[0] * (x.shape[1] // ws) * (x.shape[2] // ws), ws * ws, x.shape[3])
        E.window_reverse_layernorm = lambda win, res, g, b, eps, ws, shift: (res, win.view(res.shape))
    else:
        nn.LayerNorm.forward = _ln_fwd
        E.layernorm_window_partition, E.window_reverse_layernorm = _lnwp, _wrln


def run(wl, bs, arms):
    ctor = {"vit_b16": "vit_base_patch16_224", "swin_b": "swintransformer_base_patch4_window7_224"}[wl]
    m = getattr(models, ctor)()
    m.load_dict(seeded.fill(seeded.shapes_of(m), 1))
    m = m.to(dev).set_eval()
    x = torch.from_numpy(seeded.image_batch(32, 0)).to(dev).repeat(bs // 32, 1, 1, 1).contiguous()
    graphs = {}
    for name, (env, noln) in arms.items():
        for k in ("TLXMI_DEBUG", "TLXMI_DEBUG_COUT", "TLXMI_ATTN_DBG"):
            os.environ.pop(k, None)
        os.environ.update(env)
        skip_ln(noln)
        for _ in range(3):
            m(x)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            y = m(x)
        graphs[name] = (g, y)
        g.replay()
    skip_ln(False)
    torch.cuda.synchronize()
    ts = {k: [] for k in arms}
    for rep in range(7):
        for k in arms:
            g = graphs[k][0]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            ts[k].append(e0.elapsed_time(e1) / 10)
    base = sorted(ts["base"])[3]
    for k, t in ts.items():
        med = sorted(t)[3]
        print(f"{wl} batch {bs}  {k:34s} {med:7.3f} ms   {100 * (med - base) / base:+6.2f} %", flush=True)


if which in ("vit", "all"):
    run("vit_b16", 256, {
        "base": ({}, False),
        "qkv stores dropped": ({"TLXMI_DEBUG": "2", "TLXMI_DEBUG_COUT": "2304"}, False),
        "attention K/V staging dropped": ({"TLXMI_ATTN_DBG": "1"}, False),
        "qkv stores + K/V staging dropped": ({"TLXMI_DEBUG": "2", "TLXMI_DEBUG_COUT": "2304", "TLXMI_ATTN_DBG": "1"}, False),
        "LayerNorm launches dropped": ({}, True),
        "all three": ({"TLXMI_DEBUG": "2", "TLXMI_DEBUG_COUT": "2304", "TLXMI_ATTN_DBG": "1"}, True),
    })
if which in ("swin", "all"):
    run("swin_b", 128, {
        "base": ({}, False),
        "LayerNorm-type passes dropped": ({}, True),
        "window attention staging dropped": ({"TLXMI_ATTN_DBG": "1"}, False),
    })
