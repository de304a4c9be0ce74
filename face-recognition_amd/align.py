"""Face alignment on the GPU: utils/face_utils.py:6-71 of the reference.

``estimate_norm`` (landmarks -> 2x3 similarity matrix, utils/face_utils.py:19-53) stays host code, as in the
reference (scikit-image on the CPU): 5 points per face, vectorised over the batch in float64.  The warp
(``face_alignment``, utils/face_utils.py:56-71 = ``cv2.warpAffine(image, M, (112, 112), borderValue=0.0)``) runs as
one HIP launch for the whole batch (fr_align_faces, csrc/k_align.hip) in OpenCV's fixed-point arithmetic, and its
uint8 HWC output is exactly what ``HipBackbone.forward_u8`` takes.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence, Union

import numpy as np
import torch

from . import _lib

# utils/face_utils.py:7-16 (ArcFace template for 112x112)
REFERENCE_ALIGNMENT = np.array([[38.2946, 51.6963], [73.5318, 51.5014], [56.0252, 71.7366],
                                [41.5493, 92.3655], [70.7299, 92.2041]], dtype=np.float32)


def estimate_norm(landmarks: np.ndarray, image_size: int = 112) -> np.ndarray:
    """[n,5,2] (or [5,2]) landmark coordinates -> [n,2,3] (or [2,3]) float64 matrices that map them onto the
    template: least-squares similarity transform (Umeyama 1991), the algorithm behind
    ``SimilarityTransform.estimate`` in utils/face_utils.py:50-52, for all faces at once."""
    lm = np.asarray(landmarks, dtype=np.float64)
    single = lm.ndim == 2
    lm = lm[None] if single else lm
    if lm.shape[1:] != (5, 2):
        raise AssertionError("Landmark array must have shape (5, 2).")
    if not (image_size % 112 == 0 or image_size % 128 == 0):
        raise AssertionError("Image size must be a multiple of 112 or 128.")
    ratio = float(image_size) / (112.0 if image_size % 112 == 0 else 128.0)
    tmpl = REFERENCE_ALIGNMENT * ratio
    if image_size % 112:
        tmpl[:, 0] += 8.0 * ratio
    dst = tmpl.astype(np.float64)
    mu_s, mu_d = lm.mean(axis=1, keepdims=True), dst.mean(axis=0, keepdims=True)
    xs, xd = lm - mu_s, dst[None] - mu_d[None]
    cov = np.einsum("nki,nkj->nij", xd, xs) / 5.0                      # [n,2,2] = dst_c^T src_c / n
    U, S, Vt = np.linalg.svd(cov)
    d = np.ones((len(lm), 2))
    d[np.linalg.det(cov) < 0, 1] = -1.0
    rank = np.linalg.matrix_rank(cov)
    # rank-deficient input (collinear landmarks): the reflection is decided on det(U) det(V) instead of det(cov)
    low = rank == 1
    if low.any():
        flip = low & (np.linalg.det(U) * np.linalg.det(Vt) <= 0)
        d[low, 1] = 1.0
        d[flip, 1] = -1.0
    R = np.einsum("nij,nj,njk->nik", U, d, Vt)
    var = (xs ** 2).mean(axis=1).sum(axis=1)
    dd = np.ones((len(lm), 2))
    dd[np.linalg.det(cov) < 0, 1] = -1.0                              # the scale always uses the det(cov) sign
    scale = (S * dd).sum(axis=1) / var
    M = np.empty((len(lm), 2, 3))
    M[:, :, :2] = R * scale[:, None, None]
    M[:, :, 2] = mu_d[0][None] - scale[:, None] * np.einsum("nij,nj->ni", R, mu_s[:, 0])
    M[rank == 0] = np.nan
    return M[0] if single else M


def invert_affine(M: np.ndarray) -> np.ndarray:
    """[n,2,3] -> [n,6] float64, the inversion cv::warpAffine performs on M before warping (imgwarp.cpp)."""
    m = np.asarray(M, dtype=np.float64).reshape(-1, 6).copy()
    D = m[:, 0] * m[:, 4] - m[:, 1] * m[:, 3]
    with np.errstate(divide="ignore"):
        D = np.where(D != 0, 1.0 / D, 0.0)
    a11, a22 = m[:, 4] * D, m[:, 0] * D
    m[:, 0] = a11
    m[:, 1] *= -D
    m[:, 3] *= -D
    m[:, 4] = a22
    b1 = -m[:, 0] * m[:, 2] - m[:, 1] * m[:, 5]
    b2 = -m[:, 3] * m[:, 2] - m[:, 4] * m[:, 5]
    m[:, 2], m[:, 5] = b1, b2
    return m


def warp_faces(images: Union[torch.Tensor, Sequence[torch.Tensor]], matrices: np.ndarray, image_size: int = 112) -> torch.Tensor:
    """``cv2.warpAffine(image_i, M_i, (image_size, image_size), borderValue=0.0)`` for every face i in one launch.

    images: uint8 GPU tensors [H_i,W_i,3] (a list; sizes may differ; the same tensor may appear several times
    when one photo holds several faces) or one batched tensor [n,H,W,3].  matrices: [n,2,3].  Returns uint8
    [n,image_size,image_size,3] on the same device."""
    imgs: List[torch.Tensor] = list(images) if not isinstance(images, torch.Tensor) else [images[i] for i in range(images.shape[0])]
    M = np.asarray(matrices, dtype=np.float64).reshape(-1, 2, 3)
    if len(imgs) != len(M):
        raise ValueError(f"{len(imgs)} images for {len(M)} matrices")
    if not imgs:
        return torch.empty((0, image_size, image_size, 3), dtype=torch.uint8, device="cuda")
    dev = imgs[0].device
    for t in imgs:
        if not t.is_cuda or t.device != dev:
            raise RuntimeError("warp_faces needs all images on one GPU: the hot path has no CPU fallback")
        if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
            raise ValueError(f"expected uint8 [H,W,3] images, got {t.dtype} {tuple(t.shape)}")
    imgs = [t.contiguous() for t in imgs]
    ptrs = torch.tensor([t.data_ptr() for t in imgs], dtype=torch.int64).to(dev)
    sizes = torch.tensor([[t.shape[0], t.shape[1]] for t in imgs], dtype=torch.int32).to(dev)
    inv = torch.from_numpy(invert_affine(M)).to(dev)
    out = torch.empty((len(imgs), image_size, image_size, 3), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().fr_align_faces(ptrs.data_ptr(), sizes.data_ptr(), inv.data_ptr(), len(imgs), image_size,
                                             out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream), "fr_align_faces")
    # ptrs / sizes / inv / imgs are referenced by the launch only until it ran: the caching allocator keeps their
    # memory for this stream, so letting them go out of scope here is safe
    return out


def face_alignment(image: torch.Tensor, landmark: np.ndarray, image_size: int = 112) -> torch.Tensor:
    """utils/face_utils.py:56-71 for one face: uint8 [H,W,3] GPU image + [5,2] landmarks -> uint8 [size,size,3]."""
    return warp_faces([image], estimate_norm(landmark, image_size)[None], image_size)[0]
