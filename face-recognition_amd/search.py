"""1:N cosine gallery search over libfr_hip.so, single GPU or sharded across ranks.

The reference has no 1:N search (its only "gallery" is the JSON written by
inference.py:125-164); this module is the matcher BASELINE.json's north_star adds.  Sharding
(SURVEY.md 8(e)): gallery rows are split contiguously over the ranks of a torch.distributed
process group (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests); queries are
replicated; every rank runs the local search with GLOBAL indices and only the per-shard top-k
([Q,k] fp32 + [Q,k] int64 per rank) is exchanged by all_gather; the merge is deterministic
(descending score, ties to the lower global index), so any shard count gives the same result.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib

DIM = 512


def _stream(dev) -> int:
    return torch.cuda.current_stream(dev).cuda_stream


def shard_range(n_rows: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous row range [lo, hi) of shard ``rank``: r*G/P .. (r+1)*G/P."""
    return (n_rows * rank) // world, (n_rows * (rank + 1)) // world


def synth_rows(seed: int, row0: int, nrows: int, device, plant_seed: int = 0, n_plant: int = 0,
               n_gallery_total: int = 0, chunk: int = 1 << 22) -> torch.Tensor:
    """Synthetic L2-normalised bf16 rows [row0,row0+nrows) generated ON DEVICE by the counter-based
    generator (bit-identical to oracle/search.py): the same global gallery for any shard count."""
    out = torch.empty((nrows, DIM), dtype=torch.bfloat16, device=device)
    lib = _lib.lib()
    with torch.cuda.device(out.device):
        for lo in range(0, nrows, chunk):
            n = min(chunk, nrows - lo)
            _lib.check(lib.fr_synth_rows(out[lo:].data_ptr(), seed, row0 + lo, n, DIM, plant_seed, n_plant,
                                         n_gallery_total, _stream(out.device)), "fr_synth_rows")
    return out


def normalize_rows_bf16(x: torch.Tensor) -> torch.Tensor:
    """fp32 [n,512] embeddings -> L2-normalised bf16 rows (gallery/query format)."""
    if not x.is_cuda:
        raise RuntimeError("normalize_rows_bf16 needs a GPU tensor: the hot path has no CPU fallback")
    return torch.ops.fr_hip.l2norm_bf16(x.to(torch.float32).contiguous())          # ops.py -> fr_l2norm_bf16


def l2norm(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """fp32 rows -> (normalised fp32 rows, norms)."""
    if not x.is_cuda:
        raise RuntimeError("l2norm needs a GPU tensor: the hot path has no CPU fallback")
    x = x.to(torch.float32).contiguous()
    y = torch.empty_like(x)
    nrm = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    if x.shape[0]:
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().fr_l2norm(x.data_ptr(), y.data_ptr(), nrm.data_ptr(), x.shape[0], x.shape[1],
                                            _stream(x.device)), "fr_l2norm")
    return y, nrm


class Matcher:
    """Top-k cosine search of bf16 queries against a resident bf16 gallery (shard)."""

    def __init__(self, gallery: torch.Tensor, idx_offset: int = 0, k: int = 5) -> None:
        if not gallery.is_cuda:
            raise RuntimeError("Matcher needs the gallery on a GPU: the hot path has no CPU fallback")
        if gallery.dtype != torch.bfloat16 or gallery.dim() != 2 or gallery.shape[1] != DIM:
            raise ValueError("gallery must be bf16 [G,512] (see normalize_rows_bf16)")
        self.g = gallery.contiguous()
        self.idx_offset = int(idx_offset)
        self.k = int(k)
        self._ws: Optional[torch.Tensor] = None

    def search(self, q: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """q bf16 [Q,512] -> (scores fp32 [Q,k] descending, indices int64 [Q,k] global)."""
        if q.dtype != torch.bfloat16 or q.dim() != 2 or q.shape[1] != DIM or q.device != self.g.device:
            raise ValueError("queries must be bf16 [Q,512] on the gallery's device")
        return torch.ops.fr_hip.topk_search(q.contiguous(), self.g, self.k, self.idx_offset)   # ops.py -> fr_topk_search


def merge_topk(scores: torch.Tensor, idx: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """[P,Q,k] per-shard results -> global [Q,k] (fr_topk_merge on the GPU)."""
    if not scores.is_cuda:
        raise RuntimeError("merge_topk needs GPU tensors: the hot path has no CPU fallback")
    scores = scores.to(torch.float32).contiguous()
    idx = idx.to(torch.int64).contiguous()
    P, Q, kk = scores.shape
    if kk != k:
        raise ValueError("merge_topk: list length must equal k")
    return torch.ops.fr_hip.topk_merge(scores, idx, k)                             # ops.py -> fr_topk_merge


def all_gather_topk(scores: torch.Tensor, idx: torch.Tensor, group=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """The one exchange step of the sharded search: all_gather of [Q,k] scores and indices ->
    [P,Q,k].  With backend "nccl" this is ncclAllGather (RCCL) over xGMI; message size per rank is
    Q*k*(4+8) bytes (3.9 MB at Q = 64k, k = 5)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    Q, k = scores.shape
    gs = torch.empty((world * Q, k), dtype=scores.dtype, device=scores.device)   # concatenated along dim 0
    gi = torch.empty((world * Q, k), dtype=idx.dtype, device=idx.device)
    dist.all_gather_into_tensor(gs, scores.contiguous(), group=group)
    dist.all_gather_into_tensor(gi, idx.contiguous(), group=group)
    return gs.view(world, Q, k), gi.view(world, Q, k)


class ShardedMatcher:
    """One process per GPU; this rank holds gallery rows shard_range(G, rank, world)."""

    def __init__(self, local_gallery: torch.Tensor, n_gallery_total: int, rank: int, world: int, k: int = 5,
                 group=None) -> None:
        lo, hi = shard_range(n_gallery_total, rank, world)
        if local_gallery.shape[0] != hi - lo:
            raise ValueError(f"rank {rank} must hold rows [{lo},{hi}) of the gallery")
        self.local = Matcher(local_gallery, idx_offset=lo, k=k)
        self.k, self.world, self.group = k, world, group

    def search(self, q: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        s, i = self.local.search(q)
        import torch.distributed as dist
        if self.world == 1 and not (dist.is_available() and dist.is_initialized()):
            return s, i
        gs, gi = all_gather_topk(s, i, self.group)
        return merge_topk(gs, gi, self.k)
