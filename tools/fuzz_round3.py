"""Randomised cross-checks of the round-3 kernels against the paths they replace (tuning flavour switches): the small-block grouped
convolution vs the block-diagonal one, the filter-in-registers GEMM vs the tiled kernels, window attention with the resident
table vs the streaming form.  usage: fuzz_round3.py [cases=60] [seed=0]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tlxcv_amd import engine as E, _lib
dev = torch.device("cuda:0")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0


def close(a, b, tol, what):
    global bad
    d = (a.float() - b.float()).abs().max().item()
    s = max(1.0, b.float().abs().max().item())
    if not (d <= tol * s) or not torch.isfinite(a.float()).all():
        bad += 1
        print("MISMATCH", what, d, s, flush=True)


for case in range(n_cases):
    kind = case % 3
    if kind == 0:       # grouped conv
        cg = int(rng.choice([4, 8, 16, 32]))
        chunks = int(rng.integers(1, 5))
        C = 64 * chunks
        groups = C // cg
        stride = int(rng.choice([1, 2]))
        N, H, W = int(rng.integers(1, 4)), int(rng.integers(1, 40)), int(rng.integers(1, 100))
        act = int(rng.choice([0, 1, 2, 3, 4]))
        x = torch.randn((N, H, W, C), device=dev).half()
        w = torch.randn((C, cg, 3, 3), device=dev) * (2.0 / (cg * 9)) ** 0.5
        pk = E.PackedGroupFilter(w, groups, torch.float16)
        sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
        with _lib.tuning(TLXMI_GCONV="255"):
            a = E.group_conv2d(x, pk, stride, 1, 1, sc, sh, None, act, 0.1)
        with _lib.tuning(TLXMI_GCONV="0"):
            b = E.group_conv2d(x, pk, stride, 1, 1, sc, sh, None, act, 0.1)
        close(a, b, 4e-3, f"gconv C={C} cg={cg} s={stride} N={N} H={H} W={W} act={act}")
    elif kind == 1:     # K = 128 / 256 GEMM
        K = int(rng.choice([128, 256]))
        Cout = int(rng.choice([128, 256, 384, 512] if K == 128 else [256, 512, 768, 1024]))
        M = int(rng.integers(32768, 70000)) if K == 256 else int(rng.integers(16384, 60000))
        act = int(rng.choice([0, 1, 5, 7]))
        x = torch.randn((M, 1, 1, K), device=dev).half()
        pk = E.PackedFilter(torch.randn((Cout, K, 1, 1), device=dev) * K ** -0.5, torch.float16)
        b_ = torch.randn(Cout, device=dev) * 0.2
        sc = (torch.rand(Cout, device=dev) + 0.5) if rng.integers(0, 2) else None
        with _lib.tuning(TLXMI_WREG="3"):
            a = E.conv2d(x, pk, 1, 0, 1, sc, b_, None, act)
        with _lib.tuning(TLXMI_WREG="0"):
            b = E.conv2d(x, pk, 1, 0, 1, sc, b_, None, act)
        close(a, b, 3e-3, f"wreg M={M} K={K} N={Cout} act={act} scale={sc is not None}")
    else:               # window attention with table
        hd = int(rng.choice([32, 64]))
        heads = int(rng.choice([1, 2, 3, 4, 6, 8]))
        ws = int(rng.choice([3, 4, 5, 7, 8]))
        Ntok = ws * ws
        nW = int(rng.choice([0, 1, 4, 9]))
        imgs = int(rng.integers(1, 7))
        B = imgs * max(nW, 1)
        qkv = torch.randn((B, Ntok, 3 * heads * hd), device=dev).half()
        bias = torch.randn((heads, Ntok, Ntok), device=dev)
        mask = ((torch.rand((nW, Ntok, Ntok), device=dev) < 0.3).float() * -100.0) if nW else None
        tab = E.attention_table(bias, mask, Ntok)
        with _lib.tuning(TLXMI_WIN_STREAM="0"):
            a = E.attention_comb(qkv, heads, hd ** -0.5, tab, nW)
        with _lib.tuning(TLXMI_WIN_STREAM="1"):
            b = E.attention_comb(qkv, heads, hd ** -0.5, tab, nW)
        c = E.attention(qkv, heads, hd ** -0.5, bias, mask)
        close(a, b, 2e-3, f"attn(win) hd={hd} heads={heads} N={Ntok} nW={nW} B={B} resident vs streaming")
        close(a, c, 4e-3, f"attn(win) hd={hd} heads={heads} N={Ntok} nW={nW} B={B} vs tlxmi_attention")
    torch.cuda.synchronize()
print(f"{n_cases} cases, {bad} mismatches", flush=True)
