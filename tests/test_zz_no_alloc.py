"""fr_topk_search_noalloc (the entry point that allocates nothing: partial top-k lists in a file-scope device
array of libfr_hip.so) against fr_topk_search, through the C ABI, bit for bit -- on two calls in a row, the second
with a smaller input, so that what the first call left in the shared array is in play."""
import ctypes as C

import pytest
import torch

import face_recognition_amd as fr
from face_recognition_amd import _lib

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def _search(fn_noalloc, q, g, k, off):
    lib = _lib.lib()
    Q, G = q.shape[0], g.shape[0]
    s = torch.full((Q, k), float("nan"), dtype=torch.float32, device=DEV)
    i = torch.full((Q, k), -7, dtype=torch.int64, device=DEV)
    st = torch.cuda.current_stream(DEV).cuda_stream
    if fn_noalloc:
        _lib.check(lib.fr_topk_search_noalloc(q.data_ptr(), g.data_ptr(), Q, G, 512, k, off, s.data_ptr(), i.data_ptr(), st),
                   "fr_topk_search_noalloc")
    else:
        need = C.c_size_t()
        _lib.check(lib.fr_topk_workspace_bytes(Q, G, k, C.byref(need)), "fr_topk_workspace_bytes")
        ws = torch.empty(need.value, dtype=torch.uint8, device=DEV)
        _lib.check(lib.fr_topk_search(q.data_ptr(), g.data_ptr(), Q, G, 512, k, off, s.data_ptr(), i.data_ptr(),
                                      ws.data_ptr(), ws.numel(), st), "fr_topk_search")
    torch.cuda.synchronize(DEV)
    return s, i


def test_noalloc_search_equals_workspace_search_twice():
    lib = _lib.lib()
    torch.cuda.set_device(DEV)
    # set-up only call (what a caller does before capturing a graph): no pointers needed
    assert lib.fr_topk_search_noalloc(None, None, 0, 0, 512, 5, 0, None, None, None) == 0
    cases = [(33000, 70001, 5, 0),        # two query chunks, ragged gallery tile, many splits
             (300, 5003, 8, 1 << 33),     # smaller, k = 8, indices beyond 32 bits: runs over the first call's leftovers
             (1, 63, 1, 5)]               # smallest geometry: one query, less than a tile
    for Q, G, k, off in cases:
        g = fr.search.synth_rows(7, 0, G, DEV, plant_seed=11, n_plant=min(8, Q), n_gallery_total=G)
        q = fr.search.synth_rows(11, 0, Q, DEV)
        s1, i1 = _search(True, q, g, k, off)
        s0, i0 = _search(False, q, g, k, off)
        assert torch.equal(i1, i0), (Q, G, k)
        assert torch.equal(s1, s0), (Q, G, k)
        assert not torch.isnan(s1).any() and (i1 != -7).all()


def test_noalloc_search_makes_no_allocation_and_rejects_oversize(monkeypatch):
    lib = _lib.lib()
    torch.cuda.set_device(DEV)
    g = fr.search.synth_rows(7, 0, 20000, DEV)
    q = fr.search.synth_rows(11, 0, 512, DEV)
    s = torch.empty((512, 5), dtype=torch.float32, device=DEV)
    i = torch.empty((512, 5), dtype=torch.int64, device=DEV)
    st = torch.cuda.current_stream(DEV).cuda_stream
    assert lib.fr_topk_search_noalloc(q.data_ptr(), g.data_ptr(), 512, 20000, 512, 5, 0, s.data_ptr(), i.data_ptr(), st) == 0
    torch.cuda.synchronize(DEV)
    free0 = torch.cuda.mem_get_info(DEV)[0]
    for _ in range(3):
        assert lib.fr_topk_search_noalloc(q.data_ptr(), g.data_ptr(), 512, 20000, 512, 5, 0, s.data_ptr(), i.data_ptr(), st) == 0
    torch.cuda.synchronize(DEV)
    assert torch.cuda.mem_get_info(DEV)[0] == free0            # device memory in use did not move
    # a tuning setting that needs more partial lists than the array holds is refused with a clear error, before
    # anything is launched: 300 000 queries in ONE chunk at k = 8 need 38 MB (2 lists x 8 x 8 B per query)
    monkeypatch.setenv("FR_SEARCH_QCHUNK", str(1 << 20))
    Qb = 300_000
    big_q = fr.search.synth_rows(11, 0, Qb, DEV)
    need = C.c_size_t()
    _lib.check(lib.fr_topk_workspace_bytes(Qb, 20000, 8, C.byref(need)), "fr_topk_workspace_bytes")
    assert need.value > (32 << 20)
    sb = torch.empty((Qb, 8), dtype=torch.float32, device=DEV)
    ib = torch.empty((Qb, 8), dtype=torch.int64, device=DEV)
    rc = lib.fr_topk_search_noalloc(big_q.data_ptr(), g.data_ptr(), Qb, 20000, 512, 8, 0, sb.data_ptr(), ib.data_ptr(), st)
    assert rc == -4 and b"static scratch" in lib.fr_last_error()
    monkeypatch.delenv("FR_SEARCH_QCHUNK")
    assert lib.fr_topk_search_noalloc(q.data_ptr(), g.data_ptr(), 512, 20000, 512, 5, 0, s.data_ptr(), i.data_ptr(), st) == 0
    torch.cuda.synchronize(DEV)


def test_noalloc_search_under_hip_graph_capture():
    """The stated use of the entry point: after the Q = 0 set-up call, fr_topk_search_noalloc is captured into a
    hipGraph (torch.cuda.graph: stream capture, where hipMalloc is not allowed) and replayed twice on new queries in
    the same static buffer; each replay equals fr_topk_search on those queries bit for bit."""
    lib = _lib.lib()
    torch.cuda.set_device(DEV)
    assert lib.fr_topk_search_noalloc(None, None, 0, 0, 512, 5, 0, None, None, None) == 0
    Q, G, k = 700, 50_003, 5
    g = fr.search.synth_rows(7, 0, G, DEV, plant_seed=11, n_plant=8, n_gallery_total=G)
    q_static = fr.search.synth_rows(11, 0, Q, DEV)
    s = torch.empty((Q, k), dtype=torch.float32, device=DEV)
    i = torch.empty((Q, k), dtype=torch.int64, device=DEV)
    torch.cuda.synchronize(DEV)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        st = torch.cuda.current_stream(DEV).cuda_stream
        rc = lib.fr_topk_search_noalloc(q_static.data_ptr(), g.data_ptr(), Q, G, 512, k, 0, s.data_ptr(), i.data_ptr(), st)
    assert rc == 0, lib.fr_last_error()
    for seed in (13, 17):
        q_new = fr.search.synth_rows(seed, 0, Q, DEV)
        q_static.copy_(q_new)
        s.fill_(float("nan")); i.fill_(-7)
        graph.replay()
        torch.cuda.synchronize(DEV)
        s0, i0 = _search(False, q_new, g, k, 0)
        assert torch.equal(i, i0) and torch.equal(s, s0)
