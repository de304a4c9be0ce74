"""Generate the golden fixtures in this directory FROM THE REAL REFERENCE.

Runs only in the build container, where /root/reference exists (never on the GPU box, never from
tests).  The four backbone files are imported unmodified via a stub ``models`` package so that
``models/__init__.py`` (which needs cv2/onnxruntime) is never executed.

    python tests/golden/make_golden.py

Outputs (all small):
  state_sha256.json        sha256 of ``torch.manual_seed(0); ctor()`` state_dicts, 7 backbones
  inputs.json              sha256 of the shared uint8 input batch (regenerated from a NumPy seed)
  emb_<model>.npz          reference embeddings [8,512] fp32 (raw init, seed 0) + per-block taps
  bncal_<model>.npz        BN running stats after calibration + reference embeddings with them
  pairs.npz                64 embedding pairs, both similarity formulas, decisions at 0.35
  trained_<model>.npz      reference embeddings on the "trained-like" weight set oracle.backbones.trained_like(seed 2024)
                           (all 7 backbones; MobileNets: + the BN statistics calibrated on that weight set)
"""
import hashlib
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)

_pkg = types.ModuleType("models")
_pkg.__path__ = ["/root/reference/models"]
sys.modules["models"] = _pkg
sys.path.insert(0, "/root/reference")
from models.mobilenetv1 import MobileNetV1                      # noqa: E402
from models.mobilenetv2 import MobileNetV2                      # noqa: E402
from models.mobilenetv3 import mobilenet_v3_large, mobilenet_v3_small   # noqa: E402
from models.sphereface import sphere20, sphere36, sphere64     # noqa: E402

from oracle import backbones as ob                              # noqa: E402
from oracle import scoring as osc                               # noqa: E402

REF = {
    "sphere20": lambda: sphere20(embedding_dim=512, in_channels=3),
    "sphere36": lambda: sphere36(embedding_dim=512, in_channels=3),
    "sphere64": lambda: sphere64(embedding_dim=512, in_channels=3),
    "mobilenetv1": lambda: MobileNetV1(embedding_dim=512),
    "mobilenetv2": lambda: MobileNetV2(embedding_dim=512),
    "mobilenetv3_small": lambda: mobilenet_v3_small(embedding_dim=512),
    "mobilenetv3_large": lambda: mobilenet_v3_large(embedding_dim=512),
}


def golden_inputs():
    """uint8 [8,112,112,3]: 4 smooth synthetic faces + 4 uniform-noise images."""
    faces = osc.synthetic_faces(2, 2, seed=1234).reshape(4, 112, 112, 3)
    noise = np.random.default_rng(99).integers(0, 256, size=(4, 112, 112, 3), dtype=np.uint8)
    return np.concatenate([faces, noise], 0)


def top_blocks(net):
    if hasattr(net, "features"):
        return [f"features.{i}" for i in range(len(net.features))]
    if hasattr(net, "stage1"):
        return [f"{s}.{i}" for s in ("stage1", "stage2", "stage3") for i in range(len(getattr(net, s)))]
    return [f"layer{i}" for i in (1, 2, 3, 4)]


def tap_summary(t):
    t = t.double()
    flat = t.flatten()
    idx = torch.linspace(0, flat.numel() - 1, 16).long()
    return np.concatenate([[t.mean().item(), t.pow(2).mean().sqrt().item()], flat[idx].numpy()])


def trained_like_goldens(x):
    """The REFERENCE models on oracle.backbones.trained_like weights: per-channel log-normal
    weight scales, PReLU slopes U[0.05, 0.5], non-zero biases, BatchNorm affine parameters away from (1, 0).
    Only the embeddings (and, for the MobileNets, the BN statistics calibrated on these weights) are written;
    the weights regenerate from the seed."""
    for name, ctor in REF.items():
        torch.manual_seed(0)
        net = ctor().eval()
        net.load_state_dict(ob.trained_like(net.state_dict(), seed=2024))
        out = {}
        if name.startswith("mobilenet"):
            cal = torch.stack([osc.to_model_input(im) for im in
                               osc.synthetic_faces(12, 4, seed=78).reshape(48, 112, 112, 3)])
            ob.calibrate_bn(net, [cal[i:i + 16] for i in range(0, 48, 16)])
            stats = {k: v.numpy() for k, v in net.state_dict().items()
                     if k.endswith("running_mean") or k.endswith("running_var")}
            out["keys"] = np.array(list(stats.keys()))
            out.update({f"s{i}": v for i, v in enumerate(stats.values())})
        with torch.no_grad():
            y = net(x)
            y64 = net.double()(x.double())
        c = torch.nn.functional.cosine_similarity(y.double(), y64, dim=1)
        np.savez_compressed(os.path.join(HERE, f"trained_{name}.npz"), emb=y.numpy(), emb64=y64.numpy(), **out)
        print(name, "trained-like emb norm", y.norm(dim=1)[:2].tolist(), "fp32 vs fp64 cos", c.min().item())


def main():
    torch.set_num_threads(8)
    u8 = golden_inputs()
    x = torch.stack([osc.to_model_input(im) for im in u8])
    json.dump({"sha256": hashlib.sha256(u8.tobytes()).hexdigest(), "shape": list(u8.shape)},
              open(os.path.join(HERE, "inputs.json"), "w"))

    shas = {}
    for name, ctor in REF.items():
        torch.manual_seed(0)
        net = ctor().eval()
        shas[name] = ob.state_dict_sha256(net.state_dict())
        names = top_blocks(net)
        y, taps = ob.forward_with_taps(net, x, names)
        with torch.no_grad():
            y64 = net.double()(x.double())
        net.float()
        np.savez_compressed(os.path.join(HERE, f"emb_{name}.npz"),
                            emb=y.numpy(), emb64=y64.numpy(),
                            tap_names=np.array(names),
                            taps=np.stack([tap_summary(taps[n]) for n in names]))
        print(name, "emb norm", y.norm(dim=1)[:2].tolist())

        if name.startswith("mobilenet"):
            # BN-calibrated ("trained-like") statistics from the reference in train mode.
            cal = torch.stack([osc.to_model_input(im) for im in
                               osc.synthetic_faces(12, 4, seed=77).reshape(48, 112, 112, 3)])
            ob.calibrate_bn(net, [cal[i:i + 16] for i in range(0, 48, 16)])
            stats = {k: v.numpy() for k, v in net.state_dict().items()
                     if k.endswith("running_mean") or k.endswith("running_var")}
            with torch.no_grad():
                yc = net(x)
            np.savez_compressed(os.path.join(HERE, f"bncal_{name}.npz"), emb=yc.numpy(),
                                keys=np.array(list(stats.keys())),
                                **{f"s{i}": v for i, v in enumerate(stats.values())})
            print(name, "bn-calibrated emb norm", yc.norm(dim=1)[:2].tolist())
    json.dump(shas, open(os.path.join(HERE, "state_sha256.json"), "w"), indent=1)
    trained_like_goldens(x)

    # Pair scoring: 64 pairs of 512-d vectors; half correlated, a few straddling the threshold.
    rng = np.random.default_rng(5)
    a = rng.normal(size=(64, 512)).astype(np.float32)
    b = rng.normal(size=(64, 512)).astype(np.float32)
    for i in range(0, 32):
        b[i] = a[i] * (i / 31.0) + b[i] * (1 - i / 31.0)
    for i in range(32, 40):                       # cos ~ 0.35 +- 1e-4 steps
        u = a[i] / np.linalg.norm(a[i])
        w = b[i] - np.dot(b[i], u) * u
        w /= np.linalg.norm(w)
        c = 0.35 + (i - 35.5) * 1e-4
        b[i] = (c * u + np.sqrt(1 - c * c) * w) * 3.0
    a[40:48] *= 1e-9                              # raw-init MobileNet magnitudes
    b[40:48] *= 1e-9
    b[40:44] = a[40:44] * 1.5
    # utils/face_utils.py:84-87 (NumPy, no eps) and evaluate.py:554 (+1e-5), restated verbatim.
    sim_np = np.array([np.dot(p.ravel(), q.ravel()) / (np.linalg.norm(p.ravel()) * np.linalg.norm(q.ravel()))
                       for p, q in zip(a, b)], dtype=np.float32)
    ta, tb = torch.from_numpy(a), torch.from_numpy(b)
    sim_lfw = np.array([(f1.dot(f2) / (f1.norm() * f2.norm() + 1e-5)).item() for f1, f2 in zip(ta, tb)])
    np.savez_compressed(os.path.join(HERE, "pairs.npz"), a=a, b=b, sim_np=sim_np, sim_lfw=sim_lfw,
                        dec_np=(sim_np > 0.35), dec_lfw=(sim_lfw.astype(float) > 0.35).astype(int))
    print("pairs ok", sim_np[:4], sim_lfw[40:44])


if __name__ == "__main__":
    main()
