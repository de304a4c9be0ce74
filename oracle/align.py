"""CPU restatement of the reference's face alignment (utils/face_utils.py:6-71) -- TEST INFRASTRUCTURE ONLY.

    estimate_norm(landmark)         utils/face_utils.py:19-53   5-point similarity transform onto the ArcFace template
    face_alignment(image, landmark) utils/face_utils.py:56-71   cv2.warpAffine(image, M, (112, 112), borderValue=0.0)

Both steps live in third-party libraries that are NOT importable in this image, so this oracle is
**parity unpinned**: nothing here could be run against the reference's own code.  What is restated:

* ``skimage.transform.SimilarityTransform.estimate`` (scikit-image, unpinned in requirements.txt) = Umeyama's
  closed-form least-squares similarity (S. Umeyama, PAMI 13(4), 1991) as scikit-image's ``_umeyama`` writes
  it: float64, covariance ``dst_demean.T @ src_demean / n``, SVD, reflection fix via ``d``, scale
  ``(S @ d) / src_demean.var(axis=0).sum()``.
* ``cv2.warpAffine`` of opencv-python==4.10.0.84 (requirements.txt:2) with its defaults INTER_LINEAR,
  BORDER_CONSTANT: ``imgwarp.cpp`` inverts M in double, walks the destination in 10-bit fixed point
  (``AB_BITS``) with ``round_delta = 16``, keeps 5 fractional bits (``INTER_BITS``) and blends the 2x2 source
  pixels with 15-bit integer weights (``INTER_REMAP_COEF_BITS``), rounding with ``(sum + (1 << 14)) >> 15``.
  The 32x32 weight table holds ``(32-fx)(32-fy)*32`` etc. exactly (the products are integers), so it is
  computed here instead of tabulated; pixels outside the source are ``borderValue`` = 0.

Integer arithmetic throughout the warp: the HIP kernel (csrc/k_align.hip) must match this bit for bit.
"""
from __future__ import annotations

import numpy as np

# utils/face_utils.py:7-16
REFERENCE_ALIGNMENT = np.array([[38.2946, 51.6963], [73.5318, 51.5014], [56.0252, 71.7366],
                                [41.5493, 92.3655], [70.7299, 92.2041]], dtype=np.float32)


def umeyama(src: np.ndarray, dst: np.ndarray) -> np.ndarray:
    """3x3 homogeneous similarity transform mapping src points onto dst points (scikit-image ``_umeyama`` with
    ``estimate_scale=True``)."""
    src = np.asarray(src, dtype=np.float64)
    dst = np.asarray(dst, dtype=np.float64)
    num, dim = src.shape
    src_mean, dst_mean = src.mean(axis=0), dst.mean(axis=0)
    src_demean, dst_demean = src - src_mean, dst - dst_mean
    A = dst_demean.T @ src_demean / num
    d = np.ones((dim,), dtype=np.float64)
    if np.linalg.det(A) < 0:
        d[dim - 1] = -1
    T = np.eye(dim + 1, dtype=np.float64)
    U, S, V = np.linalg.svd(A)
    rank = np.linalg.matrix_rank(A)
    if rank == 0:
        return np.nan * T
    if rank == dim - 1:
        if np.linalg.det(U) * np.linalg.det(V) > 0:
            T[:dim, :dim] = U @ V
        else:
            s = d[dim - 1]
            d[dim - 1] = -1
            T[:dim, :dim] = U @ np.diag(d) @ V
            d[dim - 1] = s
    else:
        T[:dim, :dim] = U @ np.diag(d) @ V
    scale = 1.0 / src_demean.var(axis=0).sum() * (S @ d)
    T[:dim, dim] = dst_mean - scale * (T[:dim, :dim] @ src_mean.T)
    T[:dim, :dim] *= scale
    return T


def estimate_norm(landmark: np.ndarray, image_size: int = 112) -> np.ndarray:
    """utils/face_utils.py:19-53 -> 2x3 float64 matrix."""
    assert landmark.shape == (5, 2), "Landmark array must have shape (5, 2)."
    assert image_size % 112 == 0 or image_size % 128 == 0, "Image size must be a multiple of 112 or 128."
    if image_size % 112 == 0:
        ratio, diff_x = float(image_size) / 112.0, 0.0
    else:
        ratio, diff_x = float(image_size) / 128.0, 8.0 * float(image_size) / 128.0
    alignment = REFERENCE_ALIGNMENT * ratio             # float32, as in the reference
    alignment[:, 0] += diff_x
    return umeyama(landmark, alignment)[0:2, :]


def invert_affine(M: np.ndarray) -> np.ndarray:
    """cv::warpAffine's in-place inversion of the 2x3 matrix (imgwarp.cpp), double precision."""
    m = np.asarray(M, dtype=np.float64).reshape(6).copy()
    D = m[0] * m[4] - m[1] * m[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = m[4] * D, m[0] * D
    m[0] = A11
    m[1] *= -D
    m[3] *= -D
    m[4] = A22
    b1 = -m[0] * m[2] - m[1] * m[5]
    b2 = -m[3] * m[2] - m[4] * m[5]
    m[2], m[5] = b1, b2
    return m


def _sat_round(v: np.ndarray) -> np.ndarray:
    """saturate_cast<int>(double): cvRound (round half to even), clamped to int32."""
    return np.clip(np.rint(v), -2147483648.0, 2147483647.0).astype(np.int64)


def warp_affine_u8(image: np.ndarray, M: np.ndarray, size: int = 112) -> np.ndarray:
    """cv2.warpAffine(image, M, (size, size), borderValue=0.0) for a uint8 [H,W,C] image."""
    img = np.asarray(image)
    assert img.dtype == np.uint8 and img.ndim == 3
    H, W, C = img.shape
    m = invert_affine(M)
    AB_BITS, INTER_BITS = 10, 5
    AB_SCALE = 1 << AB_BITS
    round_delta = AB_SCALE // (1 << INTER_BITS) // 2
    xs = np.arange(size, dtype=np.float64)
    ys = np.arange(size, dtype=np.float64)
    adelta = _sat_round(m[0] * xs * AB_SCALE)
    bdelta = _sat_round(m[3] * xs * AB_SCALE)
    X0 = _sat_round((m[1] * ys + m[2]) * AB_SCALE) + round_delta
    Y0 = _sat_round((m[4] * ys + m[5]) * AB_SCALE) + round_delta
    X = (X0[:, None] + adelta[None, :]) >> (AB_BITS - INTER_BITS)
    Y = (Y0[:, None] + bdelta[None, :]) >> (AB_BITS - INTER_BITS)
    sx = np.clip(X >> INTER_BITS, -32768, 32767)
    sy = np.clip(Y >> INTER_BITS, -32768, 32767)
    fx, fy = X & 31, Y & 31
    w = [(32 - fx) * (32 - fy) * 32, fx * (32 - fy) * 32, (32 - fx) * fy * 32, fx * fy * 32]
    acc = np.zeros((size, size, C), dtype=np.int64)
    for k, (dy, dx) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
        yy, xx = sy + dy, sx + dx
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        px = img[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)].astype(np.int64)
        acc += np.where(ok[..., None], px, 0) * w[k][..., None]
    return ((acc + (1 << 14)) >> 15).astype(np.uint8)


def face_alignment(image: np.ndarray, landmark: np.ndarray, image_size: int = 112) -> np.ndarray:
    """utils/face_utils.py:56-71."""
    return warp_affine_u8(image, estimate_norm(landmark, image_size), image_size)
