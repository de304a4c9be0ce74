"""PyTorch-ROCm custom ops over the C-ABI of libfr_hip.so.

BASELINE.json's north_star asks for the hot path to be "swapped in through PyTorch-ROCm custom ops
that call HIP kernels over a thin C-ABI layer".  The ops below are registered with ``torch.library``
under the namespace ``fr_hip`` for the CUDA (= ROCm) dispatch key ONLY: calling one with CPU tensors
raises ``NotImplementedError`` from the dispatcher -- there is no CPU fallback.  Each implementation
validates dtype / shape / contiguity, allocates outputs with torch's caching allocator, and makes
exactly one C-ABI call on the current torch stream (include/fr_hip.h names the reference lines each
call replaces).

    torch.ops.fr_hip.embed(x, plan, out_c, out_h, out_w)      model(tensor)        inference.py:121-122
    torch.ops.fr_hip.embed_u8(x, plan, out_c, out_h, out_w)   transform + model    inference.py:98-106,:120-122
    torch.ops.fr_hip.cosine_pairs(a, b, eps, threshold)       similarity formulas  utils/face_utils.py:84-87, evaluate.py:554,:204-206
    torch.ops.fr_hip.l2norm_bf16(x)                           gallery/query rows   (no counterpart)
    torch.ops.fr_hip.topk_search(q, g, k, idx_offset)         1:N search           (no counterpart)
    torch.ops.fr_hip.topk_merge(scores, idx, k)               shard merge          (no counterpart)
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Tuple

import torch

from . import _lib

_NS = "fr_hip"
_defs = torch.library.Library(_NS, "DEF")
_defs.define("embed(Tensor x, int plan, int out_c, int out_h, int out_w) -> Tensor")
_defs.define("embed_u8(Tensor x, int plan, int out_c, int out_h, int out_w) -> Tensor")
_defs.define("cosine_pairs(Tensor a, Tensor b, float eps, float threshold) -> (Tensor, Tensor)")
_defs.define("l2norm_bf16(Tensor x) -> Tensor")
_defs.define("topk_search(Tensor q, Tensor g, int k, int idx_offset) -> (Tensor, Tensor)")
_defs.define("topk_merge(Tensor scores, Tensor idx, int k) -> (Tensor, Tensor)")
_impl = torch.library.Library(_NS, "IMPL", "CUDA")
_meta = torch.library.Library(_NS, "IMPL", "Meta")

# (plan handle | 0 for search, device index, stream handle) -> scratch.  Keyed by STREAM as well: two calls enqueued
# on different streams can overlap on the GPU, and a workspace shared between them (the same model on two streams,
# two Matchers) would be written by both at once.  Calls on one stream are ordered, so they share.
# Growth is bounded: at most MAX_STREAM_WORKSPACES streams per (plan, device) keep a workspace; a server that makes streams
# per request evicts the least recently used one (its memory goes back to torch's caching allocator, which hands a block
# only to later allocations of the stream it was allocated on, so kernels still queued on that stream are safe).
# ``drop_stream_workspaces()`` releases them all, or those of one stream.
_workspaces: Dict[Tuple[int, int, int], torch.Tensor] = {}
MAX_STREAM_WORKSPACES = 8


def _stream(dev) -> int:
    return torch.cuda.current_stream(dev).cuda_stream


def _workspace(key: int, dev: torch.device, nbytes: int) -> torch.Tensor:
    k = (key, dev.index, _stream(dev))
    ws = _workspaces.pop(k, None)
    if ws is None or ws.numel() < nbytes:
        ws = None
        peers = [p for p in _workspaces if p[0] == key and p[1] == dev.index]
        for p in peers[:max(0, len(peers) - (MAX_STREAM_WORKSPACES - 1))]:      # dict order = least recently used first
            del _workspaces[p]
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    _workspaces[k] = ws                                                           # (re-)inserted last = most recently used
    return ws


def drop_stream_workspaces(stream: int = -1) -> int:
    """Release the cached workspaces of one stream handle (``torch.cuda.Stream.cuda_stream``), or of every stream
    (-1).  For servers that retire streams; returns the number released."""
    ks = [k for k in _workspaces if stream == -1 or k[2] == stream]
    for k in ks:
        del _workspaces[k]
    return len(ks)


def drop_workspace(plan: int) -> None:
    """Called when a plan is destroyed."""
    for k in [k for k in _workspaces if k[0] == plan]:
        del _workspaces[k]


def _out_shape(n, c, h, w):
    return (n, c) if (h, w) == (1, 1) else (n, c, h, w)


def _embed_any(x, plan, out_c, out_h, out_w, u8):
    x = x.contiguous()
    n = x.shape[0]
    out = torch.empty(_out_shape(n, out_c, out_h, out_w), dtype=torch.float32, device=x.device)
    if n == 0:
        return out
    lib = _lib.lib()
    need = C.c_size_t()
    _lib.check(lib.fr_plan_workspace_bytes(C.c_void_p(plan), n, C.byref(need)), "fr_plan_workspace_bytes")
    ws = _workspace(plan, x.device, need.value)
    fn, name = (lib.fr_embed_u8, "fr_embed_u8") if u8 else (lib.fr_embed, "fr_embed")
    with torch.cuda.device(x.device):
        _lib.check(fn(C.c_void_p(plan), x.data_ptr(), n, out.data_ptr(), ws.data_ptr(), ws.numel(), _stream(x.device)), name)
    return out


def _embed(x, plan, out_c, out_h, out_w):
    if x.dtype != torch.float32 or x.dim() != 4:
        raise RuntimeError("fr_hip::embed: expected a float32 [N,C,H,W] tensor")
    return _embed_any(x, plan, out_c, out_h, out_w, False)


def _embed_u8(x, plan, out_c, out_h, out_w):
    if x.dtype != torch.uint8 or x.dim() != 4 or x.shape[3] != 3:
        raise RuntimeError("fr_hip::embed_u8: expected a uint8 [N,H,W,3] tensor")
    return _embed_any(x, plan, out_c, out_h, out_w, True)


def _cosine_pairs(a, b, eps, threshold):
    if a.dtype != torch.float32 or b.dtype != torch.float32 or a.dim() != 2 or a.shape != b.shape:
        raise RuntimeError("fr_hip::cosine_pairs: expected two float32 [P,d] tensors of equal shape")
    a, b = a.contiguous(), b.contiguous()
    P, d = a.shape
    sim = torch.empty(P, dtype=torch.float32, device=a.device)
    dec = torch.empty(P, dtype=torch.uint8, device=a.device)
    if P:
        with torch.cuda.device(a.device):
            _lib.check(_lib.lib().fr_cosine_pairs(a.data_ptr(), b.data_ptr(), P, d, C.c_float(eps), C.c_double(threshold),
                                                  sim.data_ptr(), dec.data_ptr(), _stream(a.device)), "fr_cosine_pairs")
    return sim, dec.bool()


def _l2norm_bf16(x):
    if x.dtype != torch.float32 or x.dim() != 2:
        raise RuntimeError("fr_hip::l2norm_bf16: expected a float32 [n,d] tensor")
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    if x.shape[0]:
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().fr_l2norm_bf16(x.data_ptr(), out.data_ptr(), x.shape[0], x.shape[1],
                                                 _stream(x.device)), "fr_l2norm_bf16")
    return out


def _topk_search(q, g, k, idx_offset):
    if q.dtype != torch.bfloat16 or g.dtype != torch.bfloat16 or q.dim() != 2 or g.dim() != 2 or q.shape[1] != g.shape[1]:
        raise RuntimeError("fr_hip::topk_search: expected bf16 [Q,d] queries and bf16 [G,d] gallery rows")
    if q.device != g.device:
        raise RuntimeError("fr_hip::topk_search: queries and gallery must be on the same device")
    q, g = q.contiguous(), g.contiguous()
    Q, G, d = q.shape[0], g.shape[0], q.shape[1]
    scores = torch.empty((Q, k), dtype=torch.float32, device=q.device)
    idx = torch.empty((Q, k), dtype=torch.int64, device=q.device)
    if Q == 0:
        return scores, idx
    if G == 0:
        return scores.fill_(float("-inf")), idx.fill_(-1)
    lib = _lib.lib()
    need = C.c_size_t()
    _lib.check(lib.fr_topk_workspace_bytes(Q, G, k, C.byref(need)), "fr_topk_workspace_bytes")
    ws = _workspace(0, q.device, need.value)
    with torch.cuda.device(q.device):
        _lib.check(lib.fr_topk_search(q.data_ptr(), g.data_ptr(), Q, G, d, k, idx_offset, scores.data_ptr(), idx.data_ptr(),
                                      ws.data_ptr(), ws.numel(), _stream(q.device)), "fr_topk_search")
    return scores, idx


def _topk_merge(scores, idx, k):
    if scores.dtype != torch.float32 or idx.dtype != torch.int64 or scores.dim() != 3 or scores.shape != idx.shape \
            or scores.shape[2] != k:
        raise RuntimeError("fr_hip::topk_merge: expected float32 / int64 [P,Q,k] tensors")
    scores, idx = scores.contiguous(), idx.contiguous()
    P, Q, _ = scores.shape
    out_s = torch.empty((Q, k), dtype=torch.float32, device=scores.device)
    out_i = torch.empty((Q, k), dtype=torch.int64, device=scores.device)
    if Q:
        with torch.cuda.device(scores.device):
            _lib.check(_lib.lib().fr_topk_merge(scores.data_ptr(), idx.data_ptr(), P, Q, k, out_s.data_ptr(),
                                                out_i.data_ptr(), _stream(scores.device)), "fr_topk_merge")
    return out_s, out_i


_impl.impl("embed", _embed)
_impl.impl("embed_u8", _embed_u8)
_impl.impl("cosine_pairs", _cosine_pairs)
_impl.impl("l2norm_bf16", _l2norm_bf16)
_impl.impl("topk_search", _topk_search)
_impl.impl("topk_merge", _topk_merge)

# shape functions (tracing / torch.export)
_meta.impl("embed", lambda x, plan, c, h, w: x.new_empty(_out_shape(x.shape[0], c, h, w), dtype=torch.float32))
_meta.impl("embed_u8", lambda x, plan, c, h, w: x.new_empty(_out_shape(x.shape[0], c, h, w), dtype=torch.float32))
_meta.impl("cosine_pairs", lambda a, b, eps, thr: (a.new_empty(a.shape[0], dtype=torch.float32),
                                                   a.new_empty(a.shape[0], dtype=torch.bool)))
_meta.impl("l2norm_bf16", lambda x: x.new_empty(x.shape, dtype=torch.bfloat16))
_meta.impl("topk_search", lambda q, g, k, off: (q.new_empty((q.shape[0], k), dtype=torch.float32),
                                                q.new_empty((q.shape[0], k), dtype=torch.int64)))
_meta.impl("topk_merge", lambda s, i, k: (s.new_empty((s.shape[1], k), dtype=torch.float32),
                                          s.new_empty((s.shape[1], k), dtype=torch.int64)))
