"""Sweep depthwise tile shapes (TH, TW, CBV) per layer shape on the GPU via the FR_DW_TILE override and
print the best few per shape.  Output feeds the tables in csrc/k_dw.hip:dw_pick_tile.
Usage (GPU): python tools/dw_probe.py <model> <batch> [bf16|f32]"""
import os, sys, itertools
os.environ["FR_DW_TABLE"] = "0"     # "default" below = the cost model of dw_pick_tile, which tools/make_dw_table.py compares against
os.environ["FR_SE_FUSE"] = "0"      # no pooling partials: their buffer is sized for the default tiling when the plan is created
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import face_recognition_amd as fr

model_name = sys.argv[1] if len(sys.argv) > 1 else "mobilenetv3_large"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = fr.get_network(model_name).to(dev).eval()
prec = sys.argv[3] if len(sys.argv) > 3 else "bf16"
model.precision = prec
ES = 32 if prec == "f32" else 16
x = (torch.rand((batch, 3, 112, 112)) * 2 - 1).to(dev)
plan = model.plan(dev)
g = fr.models._Graph(); model.lower(g)

def lds_bytes(k, s, th, tw, cbv):
    ih, iw = (th - 1) * s + k, (tw - 1) * s + k
    return ih * iw * cbv * ES + k * k * cbv * 32 + cbv * 64

def layer_ms(reps=3):
    acc = None
    for _ in range(reps):
        rows = plan.profile(x)
        acc = [r[2] for r in rows] if acc is None else [min(a, r[2]) for a, r in zip(acc, rows)]
    return acc

shapes = {}
for i, d in enumerate(g.descs):
    if d.op == 0 and d.groups == d.cin and d.groups > 1 and d.kernel in (3, 5):
        c, h, w = plan.dims[i]
        shapes.setdefault((c, h, int(d.kernel), int(d.stride)), []).append(i)
os.environ.pop("FR_DW_TILE", None)
base = layer_ms()
for (c, ho, k, s), idx in shapes.items():
    cv = c // 8
    res = []
    for th in (2, 4, 7, 8, 14, 16, 28):
        if th > ho and th - ho >= 2 and th != 2: continue
        for tw in (4, 8, 12, 16, 28, 32, 56):
            if tw - ho >= 4: continue
            for cbv in sorted({cv, (cv + 1) // 2, (cv + 2) // 3, (cv + 3) // 4, (cv + 4) // 5, (cv + 5) // 6, (cv + 7) // 8, 16, 12, 10, 8, 6, 5, 4, 3, 2, 1}):
                if cbv < 1 or cbv > min(cv, 16): continue
                if lds_bytes(k, s, th, tw, cbv) > 64 * 1024: continue
                if th * (tw // (2 if s == 2 else 4)) * cbv > 512: continue        # work items per tile (dw_pick_tile)
                os.environ["FR_DW_TILE"] = f"{c},{ho},{k},{s},{th},{tw},{cbv}"
                ms = layer_ms(2)
                res.append((sum(ms[i] for i in idx) / len(idx), th, tw, cbv, lds_bytes(k, s, th, tw, cbv)))
    os.environ.pop("FR_DW_TILE", None)
    res.sort()
    print(f"C={c} Ho={ho} k={k} s={s}: default {sum(base[i] for i in idx) / len(idx):.4f} ms; best:",
          "  ".join(f"{m:.4f}@{th}x{tw}x{cbv}({l // 1024}K)" for m, th, tw, cbv, l in res[:6]), flush=True)
