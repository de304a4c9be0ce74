"""On-disk gallery for 1:N search.

The reference keeps a gallery as ONE JSON object ``{filename: [512 floats]}`` written with indent 2
(inference.py:153-160, 1.Notebooks/Inference.ipynb cell 12): about 10 KB of text per face, parsed into
Python lists on every start -- 100 M faces would be ~1 TB of text.  This module stores what the matcher
actually consumes (csrc/k_search.hip): L2-normalised **bf16 rows**, row-major, in fixed-size shard files
that are memory-mapped and copied to HBM without parsing, plus an id table in row order.

    <dir>/manifest.json      {"format": "fr-gallery-1", "dim": 512, "dtype": "bf16", "normalized": true,
                              "rows": N, "shards": [{"file": "rows-00000.bf16", "rows": n0}, ...],
                              "ids": "ids.jsonl"}
    <dir>/rows-00000.bf16    raw little-endian bf16 [n0, dim]   (1 KiB per face at dim 512)
    <dir>/ids.jsonl          one JSON string per line, line i = id of global row i

Rows are normalised by the GPU kernel the matcher's queries go through (fr_l2norm_bf16), so a gallery
written here and a query embedded later meet in the same arithmetic.  ``import_json`` / ``export_json``
are the shims to and from the reference's format.  Rank r of a multi-GPU job loads only its contiguous
row range (``search.shard_range``), which is exactly how ``ShardedMatcher`` expects the gallery to be cut.
"""
from __future__ import annotations

import json
import os
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import search as _search

FORMAT = "fr-gallery-1"
MANIFEST = "manifest.json"
IDS = "ids.jsonl"


def _bf16_to_u16(rows: torch.Tensor) -> np.ndarray:
    return rows.contiguous().view(torch.int16).numpy().view(np.uint16)


def _u16_to_bf16(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int16)).view(torch.bfloat16)


class GalleryWriter:
    """Appends (ids, embeddings) batches to a gallery directory; ``close()`` writes the manifest.

    ``add`` takes raw fp32 embeddings **on the GPU** (what ``model(x)`` returns) and normalises them
    there; ``add_normalized`` takes rows that already are L2-normalised bf16 (any device)."""

    def __init__(self, path: str, dim: int = 512, shard_rows: int = 1 << 20) -> None:
        if dim <= 0 or shard_rows <= 0:
            raise ValueError("dim and shard_rows must be positive")
        self.path, self.dim, self.shard_rows = path, int(dim), int(shard_rows)
        os.makedirs(path, exist_ok=True)
        if os.path.exists(os.path.join(path, MANIFEST)):
            raise FileExistsError(f"{path} already holds a gallery")
        self._shards: List[Dict] = []
        self._fh = None
        self._rows = 0
        self._ids = open(os.path.join(path, IDS), "w", encoding="utf-8")
        self._closed = False

    def __enter__(self) -> "GalleryWriter":
        return self

    def __exit__(self, exc_type, exc, tb) -> None:
        if exc_type is None:
            self.close()
        else:                       # leave no manifest behind: a half-written gallery must not open
            self._abort()

    def _abort(self) -> None:
        if self._fh:
            self._fh.close()
        self._ids.close()
        self._closed = True

    def _next_shard(self) -> None:
        if self._fh:
            self._fh.close()
        name = f"rows-{len(self._shards):05d}.bf16"
        self._shards.append({"file": name, "rows": 0})
        self._fh = open(os.path.join(self.path, name), "wb")

    def add(self, ids: Sequence[str], embeddings: torch.Tensor) -> None:
        if not embeddings.is_cuda:
            raise RuntimeError("GalleryWriter.add normalises on the GPU (fr_l2norm_bf16): pass a GPU tensor, "
                               "or use add_normalized for rows that are already normalised bf16")
        if embeddings.dim() != 2 or embeddings.shape[1] != self.dim:
            raise ValueError(f"expected [n,{self.dim}] embeddings, got {tuple(embeddings.shape)}")
        self.add_normalized(ids, _search.normalize_rows_bf16(embeddings.to(torch.float32)))

    def add_normalized(self, ids: Sequence[str], rows: torch.Tensor) -> None:
        if self._closed:
            raise RuntimeError("gallery writer is closed")
        if rows.dtype != torch.bfloat16 or rows.dim() != 2 or rows.shape[1] != self.dim:
            raise ValueError(f"expected bf16 [n,{self.dim}] rows, got {rows.dtype} {tuple(rows.shape)}")
        if len(ids) != rows.shape[0]:
            raise ValueError(f"{len(ids)} ids for {rows.shape[0]} rows")
        a = _bf16_to_u16(rows.cpu())
        done = 0
        while done < len(a):
            if not self._shards or self._shards[-1]["rows"] == self.shard_rows:
                self._next_shard()
            n = min(len(a) - done, self.shard_rows - self._shards[-1]["rows"])
            a[done:done + n].astype("<u2", copy=False).tofile(self._fh)
            self._shards[-1]["rows"] += n
            done += n
        for s in ids:
            self._ids.write(json.dumps(str(s), ensure_ascii=False) + "\n")
        self._rows += len(a)

    def close(self) -> None:
        if self._closed:
            return
        if self._fh:
            self._fh.close()
        self._ids.close()
        man = {"format": FORMAT, "dim": self.dim, "dtype": "bf16", "normalized": True, "rows": self._rows,
               "shards": self._shards, "ids": IDS}
        tmp = os.path.join(self.path, MANIFEST + ".tmp")
        with open(tmp, "w") as f:
            json.dump(man, f, indent=1)
        os.replace(tmp, os.path.join(self.path, MANIFEST))      # the manifest appears last and atomically
        self._closed = True


class Gallery:
    """A gallery directory opened for reading (shards are memory-mapped on demand)."""

    def __init__(self, path: str) -> None:
        self.path = path
        with open(os.path.join(path, MANIFEST)) as f:
            man = json.load(f)
        if man.get("format") != FORMAT or man.get("dtype") != "bf16" or not man.get("normalized"):
            raise ValueError(f"{path}: not a {FORMAT} gallery")
        self.dim = int(man["dim"])
        self.rows = int(man["rows"])
        self._shards = [(s["file"], int(s["rows"])) for s in man["shards"]]
        self._ids_file = man.get("ids", IDS)
        if sum(n for _, n in self._shards) != self.rows:
            raise ValueError(f"{path}: shard rows do not add up to {self.rows}")
        for name, n in self._shards:
            size = os.path.getsize(os.path.join(path, name))
            if size != n * self.dim * 2:
                raise ValueError(f"{path}/{name}: {size} bytes, expected {n * self.dim * 2}")
        self._ids: Optional[List[str]] = None

    def __len__(self) -> int:
        return self.rows

    @property
    def ids(self) -> List[str]:
        if self._ids is None:
            with open(os.path.join(self.path, self._ids_file), encoding="utf-8") as f:
                self._ids = [json.loads(line) for line in f if line.strip()]
            if len(self._ids) != self.rows:
                raise ValueError(f"{self.path}: {len(self._ids)} ids for {self.rows} rows")
        return self._ids

    def _pieces(self, row0: int, row1: int) -> Iterable[np.ndarray]:
        start = 0
        for name, n in self._shards:
            lo, hi = max(row0, start), min(row1, start + n)
            if lo < hi:
                mm = np.memmap(os.path.join(self.path, name), dtype="<u2", mode="r", shape=(n, self.dim))
                yield mm[lo - start:hi - start]
            start += n

    def load_rows(self, device, row0: int = 0, row1: Optional[int] = None) -> torch.Tensor:
        """bf16 [row1-row0, dim] on ``device``; the range may cross shard files.  Streamed in 64-MiB pieces, so
        the host never holds more than that besides the page cache."""
        row1 = self.rows if row1 is None else row1
        if not 0 <= row0 <= row1 <= self.rows:
            raise IndexError(f"rows [{row0},{row1}) outside [0,{self.rows})")
        out = torch.empty((row1 - row0, self.dim), dtype=torch.bfloat16, device=device)
        at = 0
        step = 1 << 16                                           # 64 Ki rows = 64 MiB of host staging at a time
        for piece in self._pieces(row0, row1):
            for c0 in range(0, len(piece), step):
                part = np.array(piece[c0:c0 + step])                # private, writable copy of the mapped pages
                out[at:at + len(part)].copy_(_u16_to_bf16(part))
                at += len(part)
        return out

    def load_for_rank(self, device, rank: int = 0, world: int = 1) -> Tuple[torch.Tensor, int]:
        """(local rows, first global row) of rank ``rank`` under the matcher's contiguous sharding."""
        r0, r1 = _search.shard_range(self.rows, rank, world)
        return self.load_rows(device, r0, r1), r0

    def matcher(self, device, rank: int = 0, world: int = 1, k: int = 5, group=None):
        """``Matcher`` (world 1) or ``ShardedMatcher`` over this gallery's rows; indices it returns are
        global row numbers, i.e. positions in ``ids``."""
        rows, r0 = self.load_for_rank(device, rank, world)
        if world == 1:
            return _search.Matcher(rows, 0, k)
        return _search.ShardedMatcher(rows, self.rows, rank, world, k, group)

    def lookup(self, idx: torch.Tensor) -> List[List[Optional[str]]]:
        """ids for a [Q,k] index tensor (-1 = no match)."""
        ids = self.ids
        return [[ids[j] if j >= 0 else None for j in row] for row in idx.cpu().tolist()]


def open_gallery(path: str) -> Gallery:
    return Gallery(path)


def import_json(json_path: str, out_dir: str, device, shard_rows: int = 1 << 20, chunk: int = 65536) -> Gallery:
    """Reference gallery ``{filename: [floats]}`` (inference.py:153-160) -> gallery directory.  Row order is
    the JSON's key order (the order ``extract_batch_embeddings`` inserted them in)."""
    with open(json_path) as f:
        d = json.load(f)
    names = list(d.keys())
    dim = len(d[names[0]]) if names else 512
    with GalleryWriter(out_dir, dim, shard_rows) as w:
        for i in range(0, len(names), chunk):
            part = names[i:i + chunk]
            emb = torch.tensor([d[n] for n in part], dtype=torch.float32)
            if emb.dim() != 2 or emb.shape[1] != dim:
                raise ValueError(f"{json_path}: embeddings of unequal length")
            w.add(part, emb.to(device))
    return Gallery(out_dir)


def export_json(gallery: Gallery, json_path: str, chunk: int = 65536) -> None:
    """Gallery -> the reference's JSON format (indent 2).  The values are the stored rows, i.e. the
    L2-normalised bf16 embeddings widened to float: cosine scores against them are unchanged, the raw
    norms of the original embeddings are not recoverable."""
    ids = gallery.ids
    out: Dict[str, List[float]] = {}
    for i in range(0, gallery.rows, chunk):
        rows = gallery.load_rows("cpu", i, min(gallery.rows, i + chunk)).to(torch.float32)
        for j, r in enumerate(rows.tolist()):
            out[ids[i + j]] = r
    os.makedirs(os.path.dirname(os.path.abspath(json_path)), exist_ok=True)
    with open(json_path, "w") as f:
        json.dump(out, f, indent=2)


def build_from_images(model, device, ids: Sequence[str], images: Sequence[np.ndarray], out_dir: str,
                      batch_size: int = 1024, shard_rows: int = 1 << 20) -> Gallery:
    """Batched replacement for the per-file loop of ``extract_batch_embeddings`` (inference.py:150-153) at
    gallery scale: uint8 [112,112,3] faces -> fr_embed_u8 -> fr_l2norm_bf16 -> shard files."""
    from .inference import embed_images
    if len(ids) != len(images):
        raise ValueError(f"{len(ids)} ids for {len(images)} images")
    with GalleryWriter(out_dir, 512, shard_rows) as w:
        for i in range(0, len(images), batch_size):
            w.add(ids[i:i + batch_size], embed_images(model, device, images[i:i + batch_size], batch_size))
    return Gallery(out_dir)
