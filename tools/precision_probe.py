"""Prints min cosine of every backbone x {raw init, BN-calibrated} x precision mode against the reference's
golden embeddings (tests/golden).  Diagnostic: python tools/precision_probe.py [mode ...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import face_recognition_amd as fr  # noqa: E402
from oracle import backbones as ob  # noqa: E402
from oracle import scoring as osc  # noqa: E402

DEV = torch.device("cuda", 0)
GOLD = os.path.join(ROOT, "tests", "golden")


def cos_rows(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return (a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1) + 1e-300)


def main():
    modes = sys.argv[1:] or ["bf16", "f32"]
    faces = osc.synthetic_faces(2, 2, seed=1234).reshape(4, 112, 112, 3)
    noise = np.random.default_rng(99).integers(0, 256, size=(4, 112, 112, 3), dtype=np.uint8)
    x = torch.stack([osc.to_model_input(im) for im in np.concatenate([faces, noise], 0)]).to(DEV)
    for name in ("mobilenetv1", "mobilenetv2", "mobilenetv3_small", "mobilenetv3_large", "sphere20", "sphere36", "sphere64"):
        for cal in (False, True, "trained"):
            f = os.path.join(GOLD, {False: "emb_", True: "bncal_", "trained": "trained_"}[cal] + name + ".npz")
            if not os.path.exists(f):
                continue
            g = np.load(f)
            torch.manual_seed(0)
            m = fr.get_network(name)
            if cal:
                sd = ob.seeded(name, 0).state_dict()
                if cal == "trained":
                    sd = ob.trained_like(sd, seed=2024)
                for i, k in enumerate(g["keys"] if "keys" in g else []):
                    sd[str(k)] = torch.from_numpy(g[f"s{i}"])
                m.load_state_dict(sd)
            m = m.to(DEV).eval()
            row = []
            for p in modes:
                with torch.no_grad():
                    y = m(x, precision=p).cpu().numpy()
                c = cos_rows(y, g["emb"])
                mu = g["emb"].astype(np.float64).mean(0)
                cc = cos_rows(y - mu, g["emb"] - mu)
                row.append(f"{p}: min {c.min():.7f} mean {c.mean():.7f} centred min {cc.min():.6f}")
            print(f"{name:20s} {({False: 'raw    ', True: 'bncal  ', 'trained': 'trained'})[cal]}  " + "   ".join(row), flush=True)


if __name__ == "__main__":
    main()
