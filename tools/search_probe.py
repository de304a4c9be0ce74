"""Time fr_topk_search on synthetic rows: python tools/search_probe.py G Q [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import face_recognition_amd as fr

G = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda", 0)
g = fr.search.synth_rows(7, 0, G, dev, plant_seed=11, n_plant=64, n_gallery_total=G)
q = fr.search.synth_rows(11, 0, Q, dev)
m = fr.search.Matcher(g, k=5)
m.search(q); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    s, i = m.search(q)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"G={G} Q={Q} {dt*1e3:.2f} ms  {Q/dt:.0f} q/s  {2.0*G*512*Q/dt/1e12:.1f} TFLOP/s")
