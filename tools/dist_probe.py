"""Cost of the logits all-gather and of a barrier on whatever backend torchrun gives (rehearsal aid):
   python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/dist_probe.py
Note: two ranks sharing ONE GPU time-slice whole-chip persistent kernels (wave save/restore of 160 KB of LDS per
CU on every switch) — bench.py then reports seconds per step; that is an artefact of the rehearsal, not of the
multi-GPU path, where every rank owns its device."""
import os, time, torch, torch.distributed as dist
rank=int(os.environ["RANK"]); world=int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo", rank=rank, world_size=world)
dev=torch.device("cuda",0)
x=torch.randn(256,1000,device=dev,dtype=torch.half)
out=torch.empty(512,1000,device=dev,dtype=torch.half)
for _ in range(3): dist.all_gather_into_tensor(out,x)
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(10): dist.all_gather_into_tensor(out,x)
torch.cuda.synchronize(); print(rank,"all_gather_into_tensor ms",(time.perf_counter()-t0)*100)
t0=time.perf_counter()
for _ in range(10): dist.barrier()
print(rank,"barrier ms",(time.perf_counter()-t0)*100)
