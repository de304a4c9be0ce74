"""Rate of one (model, batch, precision) through model(x), 30 steps after warm-up.  python tools/quick_rate.py model batch prec [prec ...]"""
import sys, os, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import face_recognition_amd as fr
dev = torch.device("cuda", 0)
name, batch = sys.argv[1], int(sys.argv[2])
torch.manual_seed(0)
m = fr.get_network(name).to(dev).eval()
x = (torch.rand((batch, 3, 112, 112), generator=torch.Generator().manual_seed(0)) * 2 - 1).to(dev)
for prec in sys.argv[3:]:
    with torch.no_grad():
        for _ in range(15): m(x, precision=prec)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30): m(x, precision=prec)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 30
    print(f"{name} b{batch} {prec}: {batch / dt:9.0f} img/s  {dt * 1e3:.3f} ms", flush=True)
