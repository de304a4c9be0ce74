"""CPU emulation: which 2-3 byte ACTIVATION STORAGE formats keep the cos >= 0.9999 bar on the hard weight sets
(BN-calibrated, trained-like)?  Every activation the HIP plan materialises in HBM (output of a conv+BN+act unit, of a
project conv + residual add) is quantised by a forward hook; the arithmetic between storage points stays fp32 (the
f32 plans' split-bf16 GEMMs carry ~16-17 bits, emulated by rounding the WEIGHTS to 16 significant bits).
Uses the oracle nets (test infrastructure): a study, not product code.

    python tools/storage_format_study.py [model ...]
formats:
    fp32        nothing quantised (sanity)
    bf16        8 significant bits
    fp24        top 24 bits of the fp32 pattern, round to nearest (16 significant bits)           3   B/elem
    bfpM_gG     block floating point: G consecutive channels of a pixel share the exponent of their largest
                magnitude, M-bit two's-complement mantissas                                     ~M/8 B/elem
"""
import os, sys
import numpy as np, torch
from torch import nn
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import backbones as ob, scoring as osc

GOLD = os.path.join(ROOT, "tests", "golden")


def cos_rows(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return (a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1) + 1e-300)


def round_sig(t, bits):
    """round fp32 to `bits` significant bits (nearest, ties away) by integer arithmetic on the pattern"""
    drop = 24 - bits
    u = t.contiguous().view(torch.int32)
    u = (u + (1 << (drop - 1))) & ~((1 << drop) - 1)
    return u.view(torch.float32)


def bfp(t, mant_bits, group):
    """t [N,C,H,W] (or [N,C]); groups of `group` consecutive channels per pixel share one exponent"""
    shp = t.shape
    if t.dim() == 2:
        t = t[:, :, None, None]
    N, C, H, W = t.shape
    pad = (-C) % group
    if pad:
        t = torch.cat([t, t.new_zeros(N, pad, H, W)], 1)
    g = t.view(N, (C + pad) // group, group, H, W)
    amax = g.abs().amax(dim=2, keepdim=True).clamp_min(1e-38)
    e = torch.floor(torch.log2(amax)) + 1                      # amax < 2^e
    scale = torch.exp2(e - (mant_bits - 1))                    # q in (-2^(M-1), 2^(M-1))
    q = torch.clamp(torch.round(g / scale), -(2 ** (mant_bits - 1)) + 1, 2 ** (mant_bits - 1) - 1)
    out = (q * scale).view(N, C + pad, H, W)[:, :C]
    return out.reshape(shp)


FORMATS = {
    "fp32": lambda t: t,
    "bf16": lambda t: t.to(torch.bfloat16).to(torch.float32),
    "fp24": lambda t: round_sig(t, 16),
    "fp22(14 sig)": lambda t: round_sig(t, 14),
    "bfp16_g8": lambda t: bfp(t, 16, 8),
    "bfp16_g4": lambda t: bfp(t, 16, 4),
    "bfp15_g8": lambda t: bfp(t, 15, 8),
    "bfp16_g16": lambda t: bfp(t, 16, 16),
    "bfp14_g8": lambda t: bfp(t, 14, 8),
    "bfp12_g8": lambda t: bfp(t, 12, 8),
}


def storage_points(net):
    """modules whose output the HIP plan writes to HBM: conv->norm->act units, project conv + BN, residual blocks"""
    pts = []
    for name, m in net.named_modules():
        if isinstance(m, ob.SqueezeExcite):
            continue
        if name in set(ob.unit_names(net)) or getattr(m, "residual", False):
            pts.append((name, m))
    return pts


def run(net, x, fmt):
    q = FORMATS[fmt]
    hooks, saved = [], []
    if fmt != "fp32":
        for name, m in storage_points(net):
            hooks.append(m.register_forward_hook(lambda mod, inp, out: q(out)))
        for m in net.modules():                                 # split-bf16 weights: ~16 significant bits
            if isinstance(m, (nn.Conv2d, nn.Linear)) and getattr(m, "groups", 1) == 1:
                saved.append((m, m.weight.data.clone()))
                m.weight.data = round_sig(m.weight.data, 16)
    with torch.no_grad():
        y = net(x).numpy()
    for h in hooks: h.remove()
    for m, w in saved: m.weight.data = w
    return y


def main():
    models = [a for a in sys.argv[1:] if not a.startswith("-")] or ["mobilenetv1", "mobilenetv2", "mobilenetv3_small", "mobilenetv3_large"]
    faces = osc.synthetic_faces(2, 2, seed=1234).reshape(4, 112, 112, 3)
    noise = np.random.default_rng(99).integers(0, 256, size=(4, 112, 112, 3), dtype=np.uint8)
    x = torch.stack([osc.to_model_input(im) for im in np.concatenate([faces, noise], 0)])
    for name in models:
        for kind in ("emb", "bncal", "trained"):
            f = os.path.join(GOLD, f"{kind}_{name}.npz")
            if not os.path.exists(f):
                continue
            g = np.load(f)
            net = ob.seeded(name, 0)
            sd = net.state_dict()
            if kind == "trained":
                sd = ob.trained_like(sd, seed=2024)
            if "keys" in g:
                for i, k in enumerate(g["keys"]): sd[str(k)] = torch.from_numpy(g[f"s{i}"])
            net.load_state_dict(sd)
            net.eval()
            row = []
            for fmt in FORMATS:
                c = cos_rows(run(net, x, fmt), g["emb"])
                row.append(f"{fmt} {1 - c.min():.2e}")
            print(f"{name:18s} {kind:8s} 1-cos(min):  " + "  ".join(row), flush=True)


if __name__ == "__main__":
    main()
