"""Experiment: two half batches on two streams that own disjoint halves of the CUs (hipExtStreamCreateWithCUMask), eager launches.
An MFMA-bound launch on one half then leaves HBM to a bandwidth-bound launch on the other.  usage: cu_mask.py [workload] [batch] [split]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tlxcv_amd
from tlxcv_amd import seeded, models, engine as E

wl = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
torch.cuda.init()
tlxcv_amd.set_precision("fp16")
E.set_option("two_streams", False)
ctor = {"vit_b16": "vit_base_patch16_224", "swin_b": "swintransformer_base_patch4_window7_224"}.get(wl, wl)
m = getattr(models, ctor)()
m.load_dict(seeded.fill(seeded.shapes_of(m), 1))
m = m.to(dev).set_eval()
x = torch.from_numpy(seeded.image_batch(32, 0)).to(dev).repeat(bs // 32, 1, 1, 1).contiguous()
hip = C.CDLL("libamdhip64.so")


def masked_stream(words):
    s = C.c_void_p()
    arr = (C.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), C.c_uint32(len(words)), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


FULL = 0xFFFFFFFF
MASKS = {
    "lo/hi": ([FULL] * 4 + [0] * 4, [0] * 4 + [FULL] * 4),
    "even/odd words": ([FULL, 0] * 4, [0, FULL] * 4),
    "even/odd bits": ([0x55555555] * 8, [0xAAAAAAAA] * 8),
    "none": None,
}


def run(streams, parts=2):
    cur = torch.cuda.current_stream()
    n = bs // 2
    if streams is None:
        return m(x)
    sa, sb = streams
    sa.wait_stream(cur); sb.wait_stream(cur)
    with torch.cuda.stream(sa):
        y0 = m(x[:n])
    with torch.cuda.stream(sb):
        y1 = m(x[n:])
    cur.wait_stream(sa); cur.wait_stream(sb)
    return torch.cat((y0, y1), 0)


def timeit(streams, plan):
    from tlxcv_amd import engine as E
    hint = E.shared_plan("half" if plan == 128 else "full" if plan else None)     # per-call planning hint (TLXMI_PLAN_SHARED_*)
    hint.__enter__()
    for _ in range(3):
        run(streams)
    torch.cuda.synchronize()
    ts = []
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(10):
            run(streams)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 10 * 1e3)
    hint.__exit__()
    return sorted(ts)[2]


ref = m(x).float()
print(f"{wl} batch {bs}: one stream {timeit(None, 0):.3f} ms", flush=True)
plain = (torch.cuda.Stream(), torch.cuda.Stream())
print(f"  two plain streams, plan 128: {timeit(plain, 128):.3f} ms", flush=True)
for name, mk in MASKS.items():
    if mk is None:
        continue
    st = (masked_stream(mk[0]), masked_stream(mk[1]))
    y = run(st)
    torch.cuda.synchronize()
    err = float((y.float() - ref).abs().max())
    for plan in (128, 0):
        print(f"  masks {name}, plan {plan or 256}: {timeit(st, plan):.3f} ms   (max diff vs one stream {err:.3g})", flush=True)
