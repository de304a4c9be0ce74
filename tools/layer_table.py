"""Per-layer table of one plan: kernel, shape, time, algorithmic GB/s.  Usage (GPU):
python tools/layer_table.py <model> <batch> [bf16|f32]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import face_recognition_amd as fr

model_name = sys.argv[1] if len(sys.argv) > 1 else "mobilenetv3_large"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = fr.get_network(model_name).to(dev).eval()
model.precision = sys.argv[3] if len(sys.argv) > 3 else "bf16"
x = (torch.rand((batch, 3, 112, 112)) * 2 - 1).to(dev)
plan = model.plan(dev)
g = fr.models._Graph(); model.lower(g)
best = None
for _ in range(5):
    rows = plan.profile(x)
    best = [r[2] for r in rows] if best is None else [min(a, r[2]) for a, r in zip(best, rows)]
esz = 4 if plan.precision == "f32" else 2
tot = 0.0
for i, r in enumerate(rows):
    c, h, w = plan.dims[i]
    d = g.descs[i]
    cin = int(d.cin)
    tot += best[i]
    print(f"{i:3d} {r[1]:22s} k{int(d.kernel)} s{int(d.stride)} {cin:4d}->{c:4d} out {h:3d}x{w:<3d} {best[i]:.4f} ms", flush=True)
print(f"sum {tot:.3f} ms for batch {batch} ({plan.precision})")
