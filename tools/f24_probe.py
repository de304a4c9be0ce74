"""f24 bring-up: layer-by-layer comparison of the f24 plan against the f32 plan (unfused and fused), per backbone."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import face_recognition_amd as fr
from oracle import scoring as osc
DEV = torch.device("cuda", 0)

def cos(a, b):
    a = a.double().flatten(1); b = b.double().flatten(1)
    return ((a * b).sum(1) / (a.norm(dim=1) * b.norm(dim=1) + 1e-300)).min().item()

def main():
    names = sys.argv[1:] or ["mobilenetv3_large", "mobilenetv3_small", "mobilenetv2", "mobilenetv1"]
    faces = osc.synthetic_faces(2, 2, seed=1234).reshape(4, 112, 112, 3)
    noise = np.random.default_rng(99).integers(0, 256, size=(4, 112, 112, 3), dtype=np.uint8)
    x = torch.stack([osc.to_model_input(im) for im in np.concatenate([faces, noise], 0)]).to(DEV)
    for name in names:
        torch.manual_seed(0)
        m = fr.get_network(name).to(DEV).eval()
        nl = len(m.layer_names())
        bad = 0
        for i in range(nl):
            try:
                _, a = m.forward_tap(x, i, "f32")
                _, b = m.forward_tap(x, i, "f24")
            except Exception as e:
                print(name, i, m.layer_names()[i], "ERR", e); bad += 1; break
            torch.cuda.synchronize()
            c = cos(a, b); rel = ((a - b).abs().max() / (a.abs().max() + 1e-30)).item()
            flag = "" if (c > 0.999999 and rel < 1e-3) else "  <<<<<<"
            if flag or "-v" in sys.argv: print(f"{name} {i:3d} {m.layer_names()[i]:34s} cos {c:.9f} relmax {rel:.2e}{flag}")
            bad += bool(flag)
        with torch.no_grad():
            y32 = m(x, precision="f32"); yu = m(x, precision="f24", fuse=False); yf = m(x, precision="f24", fuse=True)
            big = m(x.repeat(40, 1, 1, 1), precision="f24")
        print(f"{name}: layers off {bad}; emb cos f24-unfused vs f32 {cos(yu, y32):.9f}; fused==unfused {torch.equal(yu, yf)} (maxdiff {(yu-yf).abs().max().item():.3e}); batch320 == batch8 {torch.equal(big[:8], yf) and torch.equal(big[312:], yf)}", flush=True)

if __name__ == "__main__":
    main()
