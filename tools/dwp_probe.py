"""Sweep tiles (TH, TW) of the fused depthwise+1x1 kernels per layer shape via the FR_DWP_TILE override and print
the best few.  Usage (GPU): python tools/dwp_probe.py <model> <batch> [bf16|f32]"""
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

def layer_ms(reps=3):
    acc = None
    for _ in range(reps):
        rows = plan.profile(x)
        acc = [r[2] for r in rows] if acc is None else [min(a, r[2]) for a, r in zip(acc, rows)]
    return acc, rows

os.environ.pop("FR_DWP_TILE", None)
base, rows = layer_ms()
shapes = {}
for i, r in enumerate(rows):
    if r[1] != "dwp_kernel":
        continue
    dwi = i - 1                                                    # the depthwise layer in front of this 1x1
    c, ho, _ = plan.dims[dwi]
    cin = int(g.descs[dwi - 1].cin) if dwi >= 1 and rows[dwi - 1][1].startswith("(fused") else 0
    shapes.setdefault((c, cin, ho, int(g.descs[dwi].kernel), int(g.descs[dwi].stride), plan.dims[i][0]), []).append(i)
for (c, cin, ho, k, s, cout), idx in shapes.items():
    res = []
    for th in (2, 4, 7, 8, 14, 16):
        for tw in (4, 8, 16, 28, 32, 56):
            os.environ["FR_DWP_TILE"] = f"{c},{cin},{th},{tw}"
            try:
                ms, _ = layer_ms(2)
            except RuntimeError:
                continue
            res.append((sum(ms[i] for i in idx) / len(idx), th, tw))
    os.environ.pop("FR_DWP_TILE", None)
    res.sort()
    print(f"C={c} cin={cin} Ho={ho} k={k} s={s} cout={cout}: default {sum(base[i] for i in idx) / len(idx):.4f} ms; best:",
          "  ".join(f"{m:.4f}@{th}x{tw}" for m, th, tw in res[:5]), flush=True)
