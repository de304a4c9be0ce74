"""CPU oracle: preprocessing, pair similarity, decisions, LFW annotation parsing, pair loop.

TEST INFRASTRUCTURE ONLY (see oracle/backbones.py header).  The reference modules holding these
few lines (inference.py, evaluate.py, utils/face_utils.py) cannot be imported in the build
container (they import torchvision / cv2 / uniface at module level), so they are
restated here from the cited lines.  The transform is "parity unpinned" at the Resize step (no
reference fixture; identity for 112x112 inputs, which is all this build feeds it).
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch


def to_model_input(img_u8_hwc: np.ndarray) -> torch.Tensor:
    """inference.py:98-106 / evaluate.py:183-187 for an already 112x112 RGB uint8 image:
    ToTensor (/255, HWC->CHW) then Normalize(0.5, 0.5).  Returns fp32 [3,112,112]."""
    if img_u8_hwc.shape != (112, 112, 3) or img_u8_hwc.dtype != np.uint8:
        raise ValueError("expected uint8 [112,112,3]")
    t = torch.from_numpy(np.ascontiguousarray(img_u8_hwc)).permute(2, 0, 1).to(torch.float32).div(255)
    return (t - 0.5) / 0.5


def compute_similarity(feat1: np.ndarray, feat2: np.ndarray) -> np.float32:
    """utils/face_utils.py:84-87 -- NumPy, no epsilon."""
    a, b = feat1.ravel(), feat2.ravel()
    return np.dot(a, b) / (np.linalg.norm(a) * np.linalg.norm(b))


def lfw_similarity(f1: torch.Tensor, f2: torch.Tensor) -> torch.Tensor:
    """evaluate.py:554 -- torch fp32 on CPU, +1e-5 in the denominator."""
    return f1.dot(f2) / (f1.norm() * f2.norm() + 1e-5)


def decide(similarities: Sequence[float], threshold: float = 0.35) -> np.ndarray:
    """evaluate.py:205-206: float64 cast, strict '>' -> int array."""
    return (np.asarray(similarities).astype(float) > threshold).astype(int)


def parse_lfw_line(root: str, line: str) -> Optional[Tuple[str, str, str]]:
    """evaluate.py:523-540: 3 tokens = same person, 4 tokens = different, anything else skipped."""
    parts = line.strip().split()
    if len(parts) == 3:
        name, i, j = parts
        return (os.path.join(root, name, f"{name}_{int(i):04d}.jpg"),
                os.path.join(root, name, f"{name}_{int(j):04d}.jpg"), "1")
    if len(parts) == 4:
        n1, i, n2, j = parts
        return (os.path.join(root, n1, f"{n1}_{int(i):04d}.jpg"),
                os.path.join(root, n2, f"{n2}_{int(j):04d}.jpg"), "0")
    return None


def read_lfw_annotations(root: str) -> Optional[List[str]]:
    """evaluate.py:391-399: first line is a header and is dropped; None if the file is missing."""
    try:
        with open(os.path.join(root, "lfw_ann.txt")) as f:
            return f.readlines()[1:]
    except FileNotFoundError:
        return None


@torch.no_grad()
def eval_pairs(net: torch.nn.Module, images: dict, pair_lines: Sequence[str], root: str = ""):
    """evaluate.py:512-572 with image decode replaced by a dict path -> uint8[112,112,3]:
    batch-1 forward per image (no caching, as the reference), evaluate.py:554 similarity,
    predictions rows [path1, path2, float, is_same_str]; returns (mean, predicts, metrics)."""
    preds = []
    for line in pair_lines:
        item = parse_lfw_line(root, line)
        if item is None:
            continue
        p1, p2, same = item
        if p1 not in images or p2 not in images:          # evaluate.py:547-549 (FileNotFoundError -> skip)
            continue
        f1 = net(to_model_input(images[p1]).unsqueeze(0)).squeeze(0)
        f2 = net(to_model_input(images[p2]).unsqueeze(0)).squeeze(0)
        preds.append([p1, p2, lfw_similarity(f1, f2).item(), same])
    if not preds:
        return 0.0, np.array([]), {}                       # evaluate.py:560-562
    preds = np.array(preds, dtype=object)
    sims = preds[:, 2].astype(float)
    return float(np.mean(sims)), preds, {"mean_similarity": np.mean(sims), "std_similarity": np.std(sims)}


def synthetic_faces(n_ids: int, per_id: int, seed: int = 1234) -> np.ndarray:
    """Synthetic 112x112 'faces' (SURVEY.md 8(d) config 1): a smooth low-frequency field per
    identity plus per-image noise.  Returns uint8 [n_ids, per_id, 112, 112, 3]."""
    rng = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.linspace(0, 1, 112), np.linspace(0, 1, 112), indexing="ij")
    out = np.empty((n_ids, per_id, 112, 112, 3), np.uint8)
    for i in range(n_ids):
        co = rng.normal(size=(3, 4, 4))
        base = np.zeros((3, 112, 112))
        for a in range(4):
            for b in range(4):
                base += co[:, a, b][:, None, None] * np.cos(np.pi * a * yy)[None] * np.cos(np.pi * b * xx)[None]
        base = base / (np.abs(base).max() + 1e-9)
        for j in range(per_id):
            img = 127.5 + 90.0 * base + rng.normal(scale=12.0, size=base.shape)
            out[i, j] = np.clip(img, 0, 255).astype(np.uint8).transpose(1, 2, 0)
    return out
