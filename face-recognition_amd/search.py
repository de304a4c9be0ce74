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

    def search_into(self, q: torch.Tensor, scores: torch.Tensor, idx: torch.Tensor) -> None:
        """fr_topk_search with caller-provided outputs (fp32 [Q,k] / int64 [Q,k] contiguous views, e.g. of a packed
        exchange message): nothing is allocated per call beyond the cached workspace."""
        Q, G = q.shape[0], self.g.shape[0]
        if Q == 0:
            return
        if G == 0:
            scores.fill_(float("-inf")); idx.fill_(-1)
            return
        assert scores.is_contiguous() and idx.is_contiguous() and scores.shape == (Q, self.k) == idx.shape
        from . import ops
        lib = _lib.lib()
        need = C.c_size_t()
        _lib.check(lib.fr_topk_workspace_bytes(Q, G, self.k, C.byref(need)), "fr_topk_workspace_bytes")
        ws = ops._workspace(0, q.device, need.value)
        with torch.cuda.device(q.device):
            _lib.check(lib.fr_topk_search(q.data_ptr(), self.g.data_ptr(), Q, G, DIM, self.k, self.idx_offset, scores.data_ptr(),
                                          idx.data_ptr(), ws.data_ptr(), ws.numel(), _stream(q.device)), "fr_topk_search")


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
    """Unpipelined form of the exchange: all_gather of [Q,k] scores and indices -> [P,Q,k] (two collectives).
    ShardedMatcher uses ``exchange_topk_chunks`` below; this one is kept for callers that already hold separate
    score / index tensors."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    Q, k = scores.shape
    gs = torch.empty((world * Q, k), dtype=scores.dtype, device=scores.device)   # concatenated along dim 0
    gi = torch.empty((world * Q, k), dtype=idx.dtype, device=idx.device)
    dist.all_gather_into_tensor(gs, scores.contiguous(), group=group)
    dist.all_gather_into_tensor(gi, idx.contiguous(), group=group)
    return gs.view(world, Q, k), gi.view(world, Q, k)


# ---- the exchange step, pipelined over query chunks ---------------------------------------------
# One packed message per rank and chunk: [Q*k fp32 scores, padded to 8 B | Q*k int64 global indices] -- 3.9 MB at
# Q = 64 k, k = 5.  The local search writes straight into the message, ONE all_gather moves it (backend "nccl" =
# ncclAllGather of RCCL over xGMI), and the merge reads the receive buffer as it is (fr_topk_merge_packed).

def packed_bytes(Q: int, k: int) -> Tuple[int, int]:
    """(offset of the index block, total bytes) of one rank's message for Q queries."""
    s_bytes = (Q * k * 4 + 7) // 8 * 8
    return s_bytes, s_bytes + Q * k * 8


def packed_views(buf: torch.Tensor, Q: int, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """The fp32 [Q,k] score block and the int64 [Q,k] index block of a message (views, no copy)."""
    off, total = packed_bytes(Q, k)
    assert buf.dtype == torch.uint8 and buf.numel() == total
    return buf[:Q * k * 4].view(torch.float32).view(Q, k), buf[off:].view(torch.int64).view(Q, k)


def exchange_topk_chunks(q: torch.Tensor, k: int, chunk: int, world: int, group, local_search, merge):
    """Sharded search of ``q`` in chunks of ``chunk`` queries with the exchange of chunk t overlapped with the
    local search of chunk t+1:

        local_search(q_chunk, scores_view, idx_view)   fills this rank's message with its shard's top-k (global indices)
        all_gather_into_tensor(..., async_op=True)     runs on the communicator's own stream (RCCL) / thread (gloo)
        merge(recv [P, bytes], Qc, out_scores, out_idx) deterministic P*k -> k merge, once the gather has landed

    Device-agnostic on purpose (the CPU test drives it over gloo with stand-in local_search / merge); on the GPU
    ``Work.wait()`` only makes the compute stream wait -- the host never blocks."""
    import torch.distributed as dist
    Q = q.shape[0]
    out_s = torch.empty((Q, k), dtype=torch.float32, device=q.device)
    out_i = torch.empty((Q, k), dtype=torch.int64, device=q.device)
    pending = None                                        # (work, recv, lo, hi) of the chunk in flight
    def finish(p):
        work, recv, lo, hi = p
        work.wait()
        merge(recv, hi - lo, out_s[lo:hi], out_i[lo:hi])
    for lo in range(0, Q, chunk):
        hi = min(Q, lo + chunk)
        _, total = packed_bytes(hi - lo, k)
        msg = torch.empty(total, dtype=torch.uint8, device=q.device)
        sv, iv = packed_views(msg, hi - lo, k)
        local_search(q[lo:hi], sv, iv)
        recv = torch.empty((world, total), dtype=torch.uint8, device=q.device)
        work = dist.all_gather_into_tensor(recv.view(-1), msg, group=group, async_op=True)
        if pending is not None:
            finish(pending)                               # chunk t-1 merges while chunk t's gather is on the wire
        pending = (work, recv, lo, hi)
    if pending is not None:
        finish(pending)
    return out_s, out_i


class ShardedMatcher:
    """One process per GPU; this rank holds gallery rows shard_range(G, rank, world)."""

    def __init__(self, local_gallery: torch.Tensor, n_gallery_total: int, rank: int, world: int, k: int = 5,
                 group=None, chunk: int = 32768) -> None:
        lo, hi = shard_range(n_gallery_total, rank, world)
        if local_gallery.shape[0] != hi - lo:
            raise ValueError(f"rank {rank} must hold rows [{lo},{hi}) of the gallery")
        self.local = Matcher(local_gallery, idx_offset=lo, k=k)
        self.k, self.world, self.group = k, world, group
        self.chunk = int(chunk)                            # queries per launch of the search kernel = per message

    def _merge(self, recv: torch.Tensor, Qc: int, out_s: torch.Tensor, out_i: torch.Tensor) -> None:
        with torch.cuda.device(recv.device):
            _lib.check(_lib.lib().fr_topk_merge_packed(recv.data_ptr(), recv.shape[0], Qc, self.k, out_s.data_ptr(),
                                                       out_i.data_ptr(), _stream(recv.device)), "fr_topk_merge_packed")

    def search(self, q: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        import torch.distributed as dist
        if self.world == 1 and not (dist.is_available() and dist.is_initialized()):
            return self.local.search(q)
        if q.dtype != torch.bfloat16 or q.dim() != 2 or q.shape[1] != DIM or q.device != self.local.g.device:
            raise ValueError("queries must be bf16 [Q,512] on the gallery's device")
        return exchange_topk_chunks(q.contiguous(), self.k, self.chunk, self.world, self.group,
                                    self.local.search_into, self._merge)
