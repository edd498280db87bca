"""Debug aid: effective per-row (a, b) of the fused LayerNorm + Linear kernels, recovered through W = identity."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tlxcv_amd import engine as E
dev = torch.device("cuda:0")
M, K = int(sys.argv[1]) if len(sys.argv) > 1 else 591, int(sys.argv[2]) if len(sys.argv) > 2 else 768
act = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rng = np.random.default_rng(0)
x = torch.from_numpy((rng.standard_normal((M, K)) * 1.5 + rng.standard_normal((M, 1)) * 0.7).astype(np.float32)).half()
w = torch.eye(K)
BIAS = 12.0 if act == 6 else 0.0          # GELU is the identity (to 1e-9) that far out: the fit stays linear
prep = E.LinearLN(w.to(dev), torch.full((K,), BIAS).to(dev), torch.ones(K).to(dev), torch.zeros(K).to(dev), torch.float16)
xf = x.float()
mean, var = xf.mean(1), xf.var(1, unbiased=False)
rstd = 1 / torch.sqrt(var + 1e-6)
for ik in (True, False):
    y = E.linear_ln(x.to(dev), prep, 1e-6, act, in_kernel=ik).float().cpu() - BIAS
    # y = a x + b per row: least squares over the columns
    xm = xf - xf.mean(1, keepdim=True)
    a = (xm * (y - y.mean(1, keepdim=True))).sum(1) / (xm * xm).sum(1)
    b = y.mean(1) - a * xf.mean(1)
    ra, rb = a / rstd, b / (-mean * rstd)
    print("in_kernel" if ik else "stats_pass", "a/rstd: min %.4f max %.4f" % (ra.min(), ra.max()), " b ratio: med %.4f" % rb.median())
    bad = ((ra - 1).abs() > 0.01).nonzero().flatten()
    print("  rows with a off by >1%:", len(bad), bad[:40].tolist())
    if len(bad):
        r = bad[0].item()
        print("  row", r, "a", a[r].item(), "rstd", rstd[r].item(), "b", b[r].item(), "want", (-mean[r] * rstd[r]).item())
        # which variance would give this a?
        print("  implied var", 1 / a[r].item() ** 2, "true var", var[r].item(), "implied mean", -b[r].item() / a[r].item(), "true mean", mean[r].item())
