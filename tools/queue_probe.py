"""Does something that takes hardware queues before the first plan slow the two-stream split down?  (Round 3: auxiliary streams per plan did,
from the third plan of a process on.)  Usage (GPU): python tools/queue_probe.py none|nccl <extra torch streams created first>
Measured with the per-device stream pool: 255-258 k img/s in every case (nccl group + collective first; 1..5 extra streams)."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import torch.distributed as dist
import face_recognition_amd as fr
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
if sys.argv[1] == "nccl":
    dist.init_process_group("nccl", rank=0, world_size=1, init_method="tcp://127.0.0.1:29533")
    t = torch.ones(4, device=dev); dist.all_reduce(t); dist.barrier(); torch.cuda.synchronize()
extra = [torch.cuda.Stream() for _ in range(int(sys.argv[2]))]
m = fr.get_network("mobilenetv3_large").to(dev).eval()
x = (torch.rand((1024, 3, 112, 112)) * 2 - 1).to(dev)
with torch.no_grad():
    for _ in range(15): m(x, precision="bf16")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): m(x, precision="bf16")
    torch.cuda.synchronize()
print(sys.argv[1:], round(1024 * 30 / (time.perf_counter() - t0)), "img/s", flush=True)
if sys.argv[1] == "nccl": dist.destroy_process_group()
