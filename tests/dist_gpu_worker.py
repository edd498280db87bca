"""One rank of the two-rank GPU rehearsal (tests/test_dist_gpu.py starts two of these as fresh processes on the one GPU
of the box; the backend is gloo because RCCL refuses two ranks on one device).  Runs the REAL engine on this rank's
shard of the resnet50_b4 golden images through tlxcv_amd.dist and writes what every rank ends up holding."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main(out_dir):
    import tlxcv_amd
    from tlxcv_amd import dist as D, models, seeded
    rank, world, _ = D.init()
    assert world == 2 and torch.distributed.get_backend() == "gloo"
    dev = torch.device("cuda", torch.cuda.current_device())
    g = np.load(os.path.join(REPO, "tests", "golden", "resnet50_b4.npz"))
    m = models.resnet50()
    m.load_dict(seeded.fill(seeded.shapes_of(m), int(g["weight_seed"])))
    m = m.to(dev).set_eval()
    x = torch.from_numpy(seeded.image_batch(4, int(g["input_seed"]))).to(dev)
    res = {}
    tlxcv_amd.set_precision("fp32")
    res["logits_global4"] = D.sharded_forward(m, x).cpu().numpy()                      # every rank holds the batch: 2 + 2
    lo, hi = D.shard_bounds(3, rank, world)
    res["logits_shard3"] = D.sharded_forward(m, x[lo:hi].contiguous(), total=3).cpu().numpy()   # per-rank shards, ragged: 2 + 1
    res["pred_global4"] = D.sharded_predict(m, x).cpu().numpy()
    tlxcv_amd.set_precision("fp16")
    res["logits_fp16"] = D.sharded_forward(m, x).float().cpu().numpy()
    # the bench loop's gather pipeline (dist.GatherPipe) over three steps of a replayed hipGraph: step i's gather comes back
    # one call later; the graph's static output is rewritten in between
    from tlxcv_amd.graph import GraphedForward
    lo, hi = D.shard_bounds(4, rank, world)
    xs = x[lo:hi].contiguous()
    gf = GraphedForward(m, xs.clone())
    pipe = D.GatherPipe()
    outs = []
    for i in range(3):
        y = gf(xs if i != 1 else xs.flip(0).contiguous())
        got = pipe.put(y)
        if got is not None:
            outs.append(got.float().cpu().numpy())
    outs.append(pipe.flush().float().cpu().numpy())
    res["pipe_steps"] = np.stack(outs)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **res)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
