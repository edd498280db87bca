"""ONE rank, the REAL backend: init_process_group("nccl", world_size=1) on the box's one GPU — loads librccl, creates a communicator — and
dist.GatherPipe(force_collective=True) around a replaying GraphedForward: the asynchronous all_gather_into_tensor runs on RCCL's own
stream next to the hipGraph replays on the caller's, exactly the pairing bench.py --gpus N times on the 8-GPU node.
Started as a fresh child process by tests/test_dist_gpu.py; writes what the rank ends up holding."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main(out_dir):
    import tlxcv_amd
    from tlxcv_amd import dist as D, models, seeded
    from tlxcv_amd.graph import GraphedForward
    os.environ["TLXMI_DIST_INIT_SINGLE"] = "1"
    rank, world, _ = D.init(backend="nccl")
    assert world == 1 and torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device())
    g = np.load(os.path.join(REPO, "tests", "golden", "resnet50_b4.npz"))
    m = models.resnet50()
    m.load_dict(seeded.fill(seeded.shapes_of(m), int(g["weight_seed"])))
    m = m.to(dev).set_eval()
    x = torch.from_numpy(seeded.image_batch(4, int(g["input_seed"]))).to(dev)
    tlxcv_amd.set_precision("fp16")
    res = {"logits_sharded": D.sharded_forward(m, x).float().cpu().numpy()}       # all_gather_logits at world 1: the local logits
    gf = GraphedForward(m, x.clone())
    pipe = D.GatherPipe(force_collective=True)
    outs = []
    for i in range(6):
        y = gf(x if i % 2 == 0 else x.flip(0).contiguous())
        got = pipe.put(y)                      # async RCCL all-gather of step i; returns step i - 1's
        if got is not None:
            outs.append(got.float().cpu().numpy())
    outs.append(pipe.flush().float().cpu().numpy())
    res["pipe_steps"] = np.stack(outs)
    # the plain collective too (all_gather_logits short-circuits at world 1: call the gather itself)
    res["gather_rows"] = D._gather_rows(gf(x).contiguous()).float().cpu().numpy()
    np.savez(os.path.join(out_dir, "rccl1.npz"), **res)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
    print("RCCL_ONE_RANK_OK", flush=True)


if __name__ == "__main__":
    main(sys.argv[1])
