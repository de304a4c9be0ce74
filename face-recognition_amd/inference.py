"""Mirror of the reference's inference.py entry points (same names, arguments, return types and
error behaviour) on top of the HIP plan.  Citations: /root/reference/inference.py."""
from __future__ import annotations

import ctypes as C
import json
import os
from pathlib import Path
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from . import models as M

_CTORS: Dict[str, Callable[[int], torch.nn.Module]] = {
    "sphere20": lambda e: M.sphere20(embedding_dim=e, in_channels=3),
    "sphere36": lambda e: M.sphere36(embedding_dim=e, in_channels=3),
    "sphere64": lambda e: M.sphere64(embedding_dim=e, in_channels=3),
    "mobilenetv1": lambda e: M.MobileNetV1(embedding_dim=e),
    "mobilenetv2": lambda e: M.MobileNetV2(embedding_dim=e),
    "mobilenetv3_small": lambda e: M.mobilenet_v3_small(embedding_dim=e),
    "mobilenetv3_large": lambda e: M.mobilenet_v3_large(embedding_dim=e),
}


def get_network(model_name: str) -> torch.nn.Module:
    """inference.py:29-52.  (Builds only the requested backbone; the reference instantiates all
    seven and returns one -- same result, ValueError for unknown names with the same message.)"""
    if model_name not in _CTORS:
        raise ValueError(f"Unsupported network '{model_name}'! Available options: {list(_CTORS.keys())}")
    return _CTORS[model_name](512)


def load_model(model_name: str, model_path: str, device: torch.device = None) -> torch.nn.Module:
    """inference.py:55-96: accepts a full ``.ckpt`` dict (key 'model', optional 'args') or a bare
    state_dict; strict load; ``.to(device).eval()``; every failure becomes RuntimeError."""
    try:
        # weights_only=False as in the reference (:60): .ckpt files pickle an argparse.Namespace.
        checkpoint = torch.load(model_path, map_location="cpu", weights_only=False)
        if "model" in checkpoint:
            embedding_dim = getattr(checkpoint["args"], "embedding_dim", 512) if "args" in checkpoint else 512
            if model_name not in _CTORS:
                raise ValueError(f"Unsupported network '{model_name}'")
            model = _CTORS[model_name](embedding_dim)
            model.load_state_dict(checkpoint["model"])
        else:
            model = get_network(model_name)
            model.load_state_dict(checkpoint)
        model.to(device).eval()
    except Exception as e:
        raise RuntimeError(f"Error loading model '{model_name}' from {model_path}: {e}")
    return model


def _transform(img) -> torch.Tensor:
    """Resize((112,112)) -> ToTensor -> Normalize(0.5, 0.5) on a PIL image (inference.py:102-106).
    torchvision resizes PIL inputs with PIL's bilinear filter, which is what is called here."""
    from PIL import Image
    if img.size != (112, 112):
        img = img.resize((112, 112), Image.BILINEAR)
    a = np.asarray(img, dtype=np.uint8)
    t = torch.from_numpy(np.ascontiguousarray(a)).permute(2, 0, 1).to(torch.float32).div(255)
    return (t - 0.5) / 0.5


def get_transform():
    """inference.py:98-106."""
    return _transform


def extract_features(model, device, img_path: str) -> np.ndarray:
    """inference.py:109-123: one image -> raw (un-normalised) embedding, np.float32 [512]."""
    from PIL import Image
    transform = get_transform()
    try:
        img = Image.open(img_path).convert("RGB")
    except Exception as e:
        raise FileNotFoundError(f"Error opening image {img_path}: {e}")
    tensor = transform(img).unsqueeze(0).to(device)
    with torch.no_grad():
        features = model(tensor).squeeze().cpu().numpy()
    return features


_STAGING: Dict[Tuple[int, bool], List[torch.Tensor]] = {}


def embed_images(model, device, images: Sequence[np.ndarray], batch_size: int = 1024) -> torch.Tensor:
    """Batched extraction: uint8 [112,112,3] arrays -> fp32 [n,512] on
    ``device``.  Same arithmetic as extract_features per image, large batches on the GPU."""
    outs: List[torch.Tensor] = []
    n = len(images)
    if n == 0:
        return torch.empty(0, 512, device=device)
    dev = torch.device(device)
    # the uint8 faces go to the GPU as they are (37.6 KB each); /255 and Normalize(0.5, 0.5) happen in the stem
    # kernel's load (fr_embed_u8), bit-identical to model(transform(img)).  Batches are gathered straight into two
    # pinned staging buffers and copied asynchronously, so that gathering batch i + 1 on the host overlaps the copy
    # and the forward of batch i (12 000 faces: the host gather, not the GPU, sets the rate of evaluate.eval).
    bs = min(batch_size, n)
    pinned = dev.type == "cuda"
    key = (bs, pinned)
    if key not in _STAGING:                                    # pinned allocations are slow: kept for the process (2 x bs x 37.6 KB)
        _STAGING.clear()
        _STAGING[key] = [torch.empty((bs, 112, 112, 3), dtype=torch.uint8, pin_memory=pinned) for _ in range(2)]
    stage = _STAGING[key] if n > bs else _STAGING[key][:1]
    done: List[Optional[torch.cuda.Event]] = [None] * len(stage)
    if pinned:
        torch.cuda.current_stream(dev).synchronize()           # an earlier call's copies out of these buffers are done
    with torch.no_grad():
        for j, i in enumerate(range(0, n, bs)):
            k = j % len(stage)
            if done[k] is not None:
                done[k].synchronize()                          # the copy that last read this buffer has finished
            m = min(bs, n - i)
            buf = stage[k][:m]
            np.stack(images[i:i + m], out=buf.numpy())
            x = buf.to(dev, non_blocking=pinned)
            if pinned:
                done[k] = torch.cuda.Event()
                done[k].record(torch.cuda.current_stream(dev))
            outs.append(model.forward_u8(x))
    return torch.cat(outs) if len(outs) > 1 else outs[0]


def _load_face_u8(img_path: str) -> np.ndarray:
    """Image.open(...).convert("RGB") + the Resize step of get_transform (inference.py:102, :115-118) -> uint8
    [112,112,3]; ToTensor and Normalize happen inside the stem kernel (fr_embed_u8)."""
    from PIL import Image
    try:
        img = Image.open(img_path).convert("RGB")
    except Exception as e:
        raise FileNotFoundError(f"Error opening image {img_path}: {e}")
    if img.size != (112, 112):
        img = img.resize((112, 112), Image.BILINEAR)
    return np.asarray(img, dtype=np.uint8)


def extract_batch_embeddings(model, device, image_folder: str, output_file: str = "assets/embeddings.json",
                             batch_size: int = 1024) -> dict:
    """inference.py:125-164: folder -> {filename: [floats]}, saved as JSON (indent 2), same file order, same
    messages; a file that cannot be read is reported and skipped (:155-156).  Unlike the reference's batch-1 loop
    the readable images are embedded in GPU batches (embed_images) -- the per-image
    arithmetic is that of extract_features, bit for bit."""
    image_extensions = (".jpg", ".jpeg", ".png", ".bmp", ".tiff")
    image_files = []
    for ext in image_extensions:
        image_files.extend(Path(image_folder).glob(f"*{ext}"))
        image_files.extend(Path(image_folder).glob(f"*{ext.upper()}"))
    embeddings_dict = {}
    print(f"Extracting embeddings from {len(image_files)} images...")
    for lo in range(0, len(image_files), batch_size):
        faces, report = [], []                     # report: (file name, row in `faces` | None, error) in FILE order
        for img_path in image_files[lo:lo + batch_size]:
            try:
                faces.append(_load_face_u8(str(img_path)))
                report.append((img_path.name, len(faces) - 1, None))
            except Exception as e:
                report.append((img_path.name, None, e))
        emb = embed_images(model, device, faces, batch_size).cpu().numpy() if faces else None
        for name, row, err in report:              # messages interleave in file order, as inference.py:150-156
            if err is not None:
                print(f"✗ Error processing {name}: {err}")
            else:
                embeddings_dict[name] = emb[row].tolist()
                print(f"✓ Processed: {name}")
    os.makedirs(os.path.dirname(output_file), exist_ok=True)
    with open(output_file, "w") as f:
        json.dump(embeddings_dict, f, indent=2)
    print(f"\nEmbeddings saved to: {output_file}")
    return embeddings_dict


def cosine_pairs(a: torch.Tensor, b: torch.Tensor, eps: float = 0.0, threshold: float = 0.35
                 ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Device pair scoring (fr_cosine_pairs): a,b fp32 [P,d] on the GPU -> (sim fp32 [P], same bool [P]).
    eps=0 is utils/face_utils.py:86; eps=1e-5 is evaluate.py:554; decision = float64(sim) > threshold."""
    if not (a.is_cuda and b.is_cuda):
        raise RuntimeError("cosine_pairs needs GPU tensors: the hot path has no CPU fallback")
    a = a.to(torch.float32).contiguous()
    b = b.to(torch.float32).contiguous()
    if a.shape != b.shape or a.dim() != 2:
        raise ValueError("cosine_pairs: expected two [P,d] tensors of equal shape")
    return torch.ops.fr_hip.cosine_pairs(a, b, float(eps), float(threshold))       # ops.py -> fr_cosine_pairs


def compute_similarity(feat1: np.ndarray, feat2: np.ndarray) -> np.float32:
    """utils/face_utils.py:74-87: host NumPy arrays in, np.float32 out -- the reference's own three NumPy lines
    (ravel, dot / (norm * norm), no epsilon).  Host glue by contract, as in the reference; the device form for
    embeddings that are already on the GPU is ``cosine_pairs`` (fr_cosine_pairs), which compare_faces and eval use."""
    feat1 = np.asarray(feat1).ravel()
    feat2 = np.asarray(feat2).ravel()
    return np.dot(feat1, feat2) / (np.linalg.norm(feat1) * np.linalg.norm(feat2))


def compare_faces(model, device, img1_path: str, img2_path: str, threshold: float = 0.35) -> tuple:
    """inference.py:167-177: (similarity np.float32, is_same np.bool_).  Both faces go through the backbone as ONE
    batch of two and are scored on the device (fr_cosine_pairs, eps = 0 as utils/face_utils.py:86): one H2D copy,
    one D2H copy of the score, instead of two forwards, two D2H copies and a host dot product."""
    faces = [_load_face_u8(img1_path), _load_face_u8(img2_path)]
    emb = embed_images(model, device, faces)
    sim, _ = cosine_pairs(emb[0:1], emb[1:2], eps=0.0, threshold=threshold)
    similarity = np.float32(sim.cpu().numpy()[0])
    is_same = similarity > threshold
    return similarity, is_same
