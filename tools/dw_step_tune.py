"""Tune depthwise tiles against the STEP, not the isolated layer (round 2: tiles that are 5-10 % faster layer by layer left the
two-stream step unchanged).  For every depthwise shape of a model, tries the best few tiles of the isolated sweep
(profiles/tuning/dw_probe_b512_<model>[_f32].txt) through FR_DW_TILE and times whole steps at batch 1024.
Usage (GPU): python tools/dw_step_tune.py <model> [bf16|f32]"""
import os, re, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import face_recognition_amd as fr

name = sys.argv[1] if len(sys.argv) > 1 else "mobilenetv3_large"
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
dev = torch.device("cuda", 0)
torch.manual_seed(0)
x = (torch.rand((1024, 3, 112, 112)) * 2 - 1).to(dev)


def step_ms(reps=120):
    torch.manual_seed(0)
    m = fr.get_network(name).to(dev).eval()
    m.precision = prec
    for _ in range(40):
        m(x)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps // 3):
            m(x)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / (reps // 3) * 1e3)
    m.invalidate()
    return best


probe = os.path.join(ROOT, "profiles", "tuning", f"dw_probe_b512_{name}{'_f32' if prec == 'f32' else ''}.txt")
os.environ.pop("FR_DW_TILE", None)
base = min(step_ms(), step_ms())
print(f"{name} {prec}: step with the current tables {base:.4f} ms", flush=True)
for line in open(probe):
    m = re.match(r"C=(\d+) Ho=(\d+) k=(\d) s=(\d): default ([\d.]+) ms; best: (.*)", line)
    if not m or float(m.group(5)) < 0.01:
        continue
    c, ho, k, s = (int(m.group(i)) for i in range(1, 5))
    res = []
    for cand in m.group(6).split()[:5]:
        mm = re.match(r"([\d.]+)@(\d+)x(\d+)x(\d+)", cand)
        th, tw, cbv = int(mm.group(2)), int(mm.group(3)), int(mm.group(4))
        os.environ["FR_DW_TILE"] = f"{c},{ho},{k},{s},{th},{tw},{cbv}"
        try:
            res.append((step_ms(), th, tw, cbv, float(mm.group(1))))
        except RuntimeError as e:          # e.g. a pooled layer whose partial-sum buffer was sized for another tiling
            res.append((float("nan"), th, tw, cbv, float(mm.group(1))))
    os.environ.pop("FR_DW_TILE", None)
    print(f"C={c} Ho={ho} k={k} s={s}: " + "  ".join(f"{t:.4f}@{th}x{tw}x{cbv}(isolated {iso:.4f})" for t, th, tw, cbv, iso in res), flush=True)
