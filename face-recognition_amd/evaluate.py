"""Mirror of the reference's LFW pair-scoring path: evaluate.eval (LFW branch), extract_deep_features
and the threshold decision.  Citations: /root/reference/evaluate.py.

Unlike the reference's batch-1 loop (two forwards per pair, no caching, :551-552) this driver
de-duplicates the images, embeds them in large GPU batches and scores all pairs with one
fr_cosine_pairs launch -- while returning the identical
``(mean_similarity, predictions, metrics)`` triple.  Metric/plot helpers (:197-356), non-LFW
loaders and RetinaFace validation are out of scope.
"""
from __future__ import annotations

import os
from typing import Dict, List, Tuple

import numpy as np
import torch

from .inference import _transform, cosine_pairs, embed_images


def extract_deep_features(model, img, device):
    """evaluate.py:179-194: PIL image -> CPU torch tensor [512]."""
    img_tensor = _transform(img).unsqueeze(0).to(device)
    with torch.no_grad():
        features = model(img_tensor)
    return features.squeeze(0).cpu()


def compute_metrics_from_predictions(predictions, threshold=0.35):
    """evaluate.py:197-251 restricted to the NumPy part: the decision rule (:204-206) and the
    confusion counts / accuracy derived from it (sklearn-free)."""
    if len(predictions) == 0:
        return {}
    y_true = predictions[:, 3].astype(int)
    similarities = predictions[:, 2].astype(float)
    y_pred = (similarities > threshold).astype(int)
    tp = int(((y_pred == 1) & (y_true == 1)).sum()); tn = int(((y_pred == 0) & (y_true == 0)).sum())
    fp = int(((y_pred == 1) & (y_true == 0)).sum()); fn = int(((y_pred == 0) & (y_true == 1)).sum())
    precision = tp / (tp + fp) if tp + fp else 0.0
    recall = tp / (tp + fn) if tp + fn else 0.0
    thresholds = np.linspace(similarities.min(), similarities.max(), 100)
    accs = [float((((similarities > t).astype(int)) == y_true).mean()) for t in thresholds]
    best = int(np.argmax(accs))
    return {
        "accuracy": (tp + tn) / len(y_true), "precision": precision, "recall": recall,
        "f1": 2 * precision * recall / (precision + recall) if precision + recall else 0.0,
        "confusion_matrix": np.array([[tn, fp], [fn, tp]]),
        "mean_similarity": np.mean(similarities), "std_similarity": np.std(similarities),
        "min_similarity": np.min(similarities), "max_similarity": np.max(similarities),
        "median_similarity": np.median(similarities),
        "best_threshold": thresholds[best], "best_accuracy": accs[best],
        "true_negatives": tn, "false_positives": fp, "false_negatives": fn, "true_positives": tp,
    }


def _lfw_pairs(root: str, pair_lines) -> List[Tuple[str, str, str]]:
    """evaluate.py:523-540."""
    out = []
    for line in pair_lines:
        parts = line.strip().split()
        if len(parts) == 3:
            name, i, j = parts
            out.append((os.path.join(root, name, f"{name}_{int(i):04d}.jpg"),
                        os.path.join(root, name, f"{name}_{int(j):04d}.jpg"), "1"))
        elif len(parts) == 4:
            n1, i, n2, j = parts
            out.append((os.path.join(root, n1, f"{n1}_{int(i):04d}.jpg"),
                        os.path.join(root, n2, f"{n2}_{int(j):04d}.jpg"), "0"))
    return out


def score_pairs(model, device, pairs, loader, threshold: float = 0.35, batch_size: int = 1024):
    """pairs: [(path1, path2, is_same_str)]; loader(path) -> uint8 [112,112,3] or raises
    FileNotFoundError.  Returns the reference's predictions object array [P,4]
    (path1, path2, float similarity, is_same) with evaluate.py:554 scoring (+1e-5)."""
    cache: Dict[str, int] = {}
    images: List[np.ndarray] = []
    kept = []
    skipped = 0
    for p1, p2, same in pairs:
        try:
            for p in (p1, p2):
                if p not in cache:
                    img = loader(p)
                    cache[p] = len(images)
                    images.append(img)
        except FileNotFoundError:
            skipped += 1                                   # evaluate.py:547-549
            continue
        kept.append((p1, p2, same))
    if skipped > 0:
        print(f"Skipped {skipped} pairs (images not found)")
    if not kept:
        return np.array([])
    emb = embed_images(model, device, images, batch_size)
    ia = torch.tensor([cache[p1] for p1, _, _ in kept], device=emb.device)
    ib = torch.tensor([cache[p2] for _, p2, _ in kept], device=emb.device)
    sim, _ = cosine_pairs(emb[ia], emb[ib], eps=1e-5, threshold=threshold)
    sims = sim.cpu().numpy()
    return np.array([[p1, p2, float(s), same] for (p1, p2, same), s in zip(kept, sims)], dtype=object)


def _load_image(path: str) -> np.ndarray:
    from PIL import Image
    img = Image.open(path).convert("RGB")                   # raises FileNotFoundError like :545-546
    if img.size != (112, 112):
        img = img.resize((112, 112), Image.BILINEAR)
    return np.asarray(img, dtype=np.uint8)


def eval(model, model_path=None, device=None, val_dataset="lfw", val_root="data/lfw/val",
         compute_full_metrics=False, save_metrics_path=None, threshold=0.35, face_validator=None,
         no_face_policy="exclude", use_landmarks=False, landmark_cache_dir="landmark_cache",
         image_loader=None):
    """evaluate.py:359-589, LFW branch.  ``image_loader`` (extra, optional) replaces image decode
    for synthetic data: path -> uint8 [112,112,3]."""
    if device is None:
        device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    if model_path is not None:
        state_dict = torch.load(model_path, map_location="cpu")
        model.load_state_dict(state_dict)
    model.to(device).eval()
    if val_dataset != "lfw":
        raise ValueError(f"Unsupported validation dataset: {val_dataset}.")
    if face_validator is not None:
        raise NotImplementedError("RetinaFace pair validation is out of scope of the hot path (SURVEY.md 2, #17)")
    ann_file = os.path.join(val_root, "lfw_ann.txt")
    try:
        with open(ann_file) as f:
            pair_lines = f.readlines()[1:]                   # header dropped, :395-396
    except FileNotFoundError:
        print(f"ERROR: Annotation file 'lfw_ann.txt' not found in '{val_root}'.")
        return 0.0, np.array([]), {}
    predicts = score_pairs(model, device, _lfw_pairs(val_root, pair_lines), image_loader or _load_image, threshold)
    if len(predicts) == 0:
        print("Warning: No valid pairs were processed.")
        return 0.0, np.array([]), {}
    similarities = predicts[:, 2].astype(float)
    mean_similarity = np.mean(similarities)
    metrics = {"mean_similarity": mean_similarity, "std_similarity": np.std(similarities)}
    if compute_full_metrics:
        metrics.update(compute_metrics_from_predictions(predicts, threshold))
    if save_metrics_path is not None:
        # evaluate.py:578-586 writes ROC / confusion-matrix plots there (sklearn + matplotlib reporting, SURVEY.md 2
        # #7: out of scope of the hot path) -- say so instead of silently ignoring the argument.
        import warnings
        warnings.warn(f"save_metrics_path={save_metrics_path!r}: metric plots are not written by this implementation "
                      "(out of scope of the embed-and-match path); the returned metrics dict is complete", RuntimeWarning,
                      stacklevel=2)
    return mean_similarity, predicts, metrics
