"""CPU oracle: synthetic gallery generator and definitional 1:N top-k cosine search.

TEST INFRASTRUCTURE ONLY (see oracle/backbones.py header).

The reference has NO 1:N search: parity is *unpinned by the
reference*; the oracle is the definitional ``normalize(q) @ normalize(g).T -> top-k`` in fp32/fp64
(the eval branch of utils/metrics.py:18-28 is the closest reference arithmetic).

Synthetic rows come from a counter-based generator designed to be
bit-reproducible between NumPy and the HIP kernel: every element is an *integer*
    v(seed,row,col) = sum of the four 16-bit fields of splitmix64(seed, row*dim+col) - 2*65535
(Irwin-Hall, approx. normal, sigma ~ 37837), the row's sum of squares is an exact int64, the
scale 1/sqrt(sumsq) is computed in IEEE double and rounded to fp32 once, and each element is
fp32(v) * scale rounded to bf16 (RNE).  Planted matches: gallery row ``plant_row(t)`` holds
3*v_query(t) + v_gallery(row), so query t has a known nearest neighbour (cos ~ 0.95).
"""
from __future__ import annotations

from typing import Tuple

import numpy as np

M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
GOLD = np.uint64(0x9E3779B97F4A7C15)


def splitmix64(seed: int, idx: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (idx.astype(np.uint64) + np.uint64(1)) * GOLD
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def int_rows(seed: int, row0: int, nrows: int, dim: int = 512) -> np.ndarray:
    """Integer vectors v[row0:row0+nrows, :dim] (int64)."""
    idx = (np.arange(row0, row0 + nrows, dtype=np.uint64)[:, None] * np.uint64(dim)
           + np.arange(dim, dtype=np.uint64)[None, :])
    z = splitmix64(seed, idx)
    s = ((z & np.uint64(0xFFFF)) + ((z >> np.uint64(16)) & np.uint64(0xFFFF))
         + ((z >> np.uint64(32)) & np.uint64(0xFFFF)) + (z >> np.uint64(48)))
    return s.astype(np.int64) - 2 * 65535


def plant_row(t: int, n_gallery: int, n_plant: int) -> int:
    """Gallery row that holds the planted match of query t (t < n_plant); rows are distinct."""
    return (t * (n_gallery // n_plant) + 17) % n_gallery


def f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    return ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)).astype(np.uint16)


def bf16_bits_to_f32(b: np.ndarray) -> np.ndarray:
    return (b.astype(np.uint32) << np.uint32(16)).view(np.float32)


def _normalise_to_bf16(v: np.ndarray) -> np.ndarray:
    sumsq = (v * v).sum(axis=1)                                   # exact int64
    scale = (1.0 / np.sqrt(sumsq.astype(np.float64))).astype(np.float32)
    return f32_to_bf16_bits(v.astype(np.float32) * scale[:, None])


def query_rows(seed_q: int, row0: int, nrows: int, dim: int = 512) -> np.ndarray:
    """bf16 bit patterns (uint16) of L2-normalised synthetic queries."""
    return _normalise_to_bf16(int_rows(seed_q, row0, nrows, dim))


def gallery_rows(seed_g: int, seed_q: int, row0: int, nrows: int, n_gallery: int, n_plant: int,
                 dim: int = 512) -> np.ndarray:
    """bf16 bit patterns (uint16) of gallery rows [row0, row0+nrows) of the global gallery."""
    v = int_rows(seed_g, row0, nrows, dim)
    if n_plant > 0:
        step = n_gallery // n_plant
        for r in range(row0, row0 + nrows):
            d = (r - 17) % n_gallery
            if d % step == 0 and d // step < n_plant:
                v[r - row0] += 3 * int_rows(seed_q, d // step, 1, dim)[0]
    return _normalise_to_bf16(v)


def topk_cosine(q: np.ndarray, g: np.ndarray, k: int = 5, idx_offset: int = 0,
                dtype=np.float64) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Definitional search.  q [Q,d], g [G,d] real-valued (already normalised or not: both are
    normalised here).  Returns (scores [Q,k] descending, indices [Q,k] int64 with ties broken by
    the lower index, gap [Q] = score_k - score_(k+1), +inf if G == k)."""
    q = np.asarray(q, dtype)
    g = np.asarray(g, dtype)
    q = q / np.linalg.norm(q, axis=1, keepdims=True)
    g = g / np.linalg.norm(g, axis=1, keepdims=True)
    s = q @ g.T
    kk = min(k + 1, s.shape[1])
    order = np.lexsort((np.broadcast_to(np.arange(s.shape[1]), s.shape), -s), axis=1)[:, :kk]
    top = np.take_along_axis(s, order, axis=1)
    gap = top[:, k - 1] - top[:, k] if kk > k else np.full(s.shape[0], np.inf)
    return top[:, :k], order[:, :k].astype(np.int64) + idx_offset, gap


def merge_topk(scores: np.ndarray, idx: np.ndarray, k: int = 5) -> Tuple[np.ndarray, np.ndarray]:
    """Merge per-shard results [P,Q,k] -> [Q,k]; descending score, ties to the lower global index."""
    P, Q, kk = scores.shape
    s = scores.transpose(1, 0, 2).reshape(Q, P * kk)
    i = idx.transpose(1, 0, 2).reshape(Q, P * kk)
    order = np.lexsort((i, -s), axis=1)[:, :k]
    return np.take_along_axis(s, order, axis=1), np.take_along_axis(i, order, axis=1)


def topk_cosine_fp32_chunked(q, g, k: int = 5, chunk: int = 65536):
    """The CPU baseline of the matcher: fp32 ``Q @ G_chunk.T`` + ``torch.topk(k)`` per gallery
    chunk with a running merge, all host cores (PyTorch CPU / oneDNN).  q [Q,d], g [G,d] torch fp32, rows already
    L2-normalised.  Same definition as topk_cosine (which is float64 NumPy with an explicit tie order and is the
    checker); this one is the form a CPU user of the reference would write for speed.  Returns (scores, indices)."""
    import torch
    Q = q.shape[0]
    best_s = torch.full((Q, k), float("-inf"))
    best_i = torch.full((Q, k), -1, dtype=torch.int64)
    for lo in range(0, g.shape[0], chunk):
        s = q @ g[lo:lo + chunk].T
        cs, ci = torch.topk(s, min(k, s.shape[1]), dim=1)
        ms = torch.cat([best_s, cs], dim=1)
        mi = torch.cat([best_i, ci + lo], dim=1)
        best_s, sel = torch.topk(ms, k, dim=1)
        best_i = torch.gather(mi, 1, sel)
    return best_s, best_i
