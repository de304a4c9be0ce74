"""CPU emulation: which tensors have to be kept at more than bf16 for the cos >= 0.9999 bar on BN-calibrated weights?
Every dense conv / linear gets its INPUT and/or WEIGHT rounded to bf16 according to a policy (fp32 accumulate), and the
embedding is compared with the golden.  Uses the oracle nets (test infrastructure): a study, not product code.
python tools/mixed_precision_study.py [model ...]"""
import os, sys
import numpy as np, torch
from torch import nn
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import backbones as ob, scoring as osc

GOLD = os.path.join(ROOT, "tests", "golden")
bf = lambda t: t.to(torch.bfloat16).to(torch.float32)


def cos_rows(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return (a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1) + 1e-300)


def run(net, x, policy):
    """policy(name, module) -> (round_input, round_weight)"""
    hooks, saved = [], []
    for name, m in net.named_modules():
        if isinstance(m, (nn.Conv2d, nn.Linear)):
            ri, rw = policy(name, m)
            if rw:
                saved.append((m, m.weight.data.clone()))
                m.weight.data = bf(m.weight.data)
            if ri:
                hooks.append(m.register_forward_pre_hook(lambda mod, inp: (bf(inp[0]),)))
    with torch.no_grad():
        y = net(x).numpy()
    for h in hooks: h.remove()
    for m, w in saved: m.weight.data = w
    return y


def kind(name, m):
    if isinstance(m, nn.Linear): return "linear"
    if m.groups > 1: return "dw"
    if m.kernel_size[0] == 3: return "stem"
    if "fc1" in name or "fc2" in name: return "se"
    return "expand" if m.out_channels > m.in_channels else "project"


def main():
    models = sys.argv[1:] or ["mobilenetv1", "mobilenetv2", "mobilenetv3_small", "mobilenetv3_large"]
    faces = osc.synthetic_faces(2, 2, seed=1234).reshape(4, 112, 112, 3)
    noise = np.random.default_rng(99).integers(0, 256, size=(4, 112, 112, 3), dtype=np.uint8)
    x = torch.stack([osc.to_model_input(im) for im in np.concatenate([faces, noise], 0)])
    P = {
        "all bf16 (in+w)": lambda n, m: (kind(n, m) != "se", kind(n, m) != "se"),
        "weights fp32, acts bf16": lambda n, m: (kind(n, m) != "se", False),
        "acts fp32, weights bf16": lambda n, m: (False, kind(n, m) != "se"),
        "narrow (expand/stem input) fp32": lambda n, m: (kind(n, m) not in ("se", "expand", "stem"), False),
        "wide (project input) fp32": lambda n, m: (kind(n, m) not in ("se", "project", "linear"), False),
        "dw input fp32": lambda n, m: (kind(n, m) not in ("se", "dw"), False),
        "dw+project input fp32 (expand in bf16)": lambda n, m: (kind(n, m) in ("expand", "stem"), False),
        "only dw input bf16": lambda n, m: (kind(n, m) == "dw", False),
        "only project input bf16": lambda n, m: (kind(n, m) in ("project", "linear"), False),
        "only expand input bf16": lambda n, m: (kind(n, m) in ("expand", "stem"), False),
    }
    for name in models:
        for cal in (False, True):
            g = np.load(os.path.join(GOLD, ("bncal_" if cal else "emb_") + name + ".npz"))
            net = ob.seeded(name, 0)
            if cal:
                sd = net.state_dict()
                for i, k in enumerate(g["keys"]): sd[str(k)] = torch.from_numpy(g[f"s{i}"])
                net.load_state_dict(sd)
            net.eval()
            for pn, pol in P.items():
                c = cos_rows(run(net, x, pol), g["emb"])
                print(f"{name:18s} {'bncal' if cal else 'raw  '} {pn:42s} min {c.min():.7f}", flush=True)


if __name__ == "__main__":
    main()
