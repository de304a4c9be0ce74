"""CPU: host logic of the product package and the C-ABI library (no compute calls: no GPU here)."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

import face_recognition_amd as fr
from face_recognition_amd import _lib
from oracle import backbones as ob

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "fr_hip.h")).read()
    declared = set(re.findall(r"\b(fr_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"fr_layer_desc"}
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    lib = _lib.lib()
    for name in declared:
        assert hasattr(lib, name), name
    assert ctypes.sizeof(_lib.LayerDesc) == 10 * 4 + 6 * 8 + 8 + 3 * 8   # matches the C struct layout


@pytest.mark.parametrize("name", ob.MODEL_NAMES)
def test_product_random_init_equals_reference(name, golden_dir):
    sha = json.load(open(os.path.join(golden_dir, "state_sha256.json")))
    torch.manual_seed(0)
    m = fr.get_network(name)
    assert ob.state_dict_sha256(m.state_dict()) == sha[name]


@pytest.mark.parametrize("name", ["mobilenetv1", "mobilenetv2", "mobilenetv3_small", "mobilenetv3_large", "sphere20"])
def test_lowering_covers_every_unit(name):
    m = fr.get_network(name).eval()
    names = m.layer_names()
    units = ob.unit_names(ob.build(name))
    assert names[:len(units)] == units or set(units) <= set(names)
    assert len(set(names)) == len(names)


def test_checkpoint_roundtrip_and_error_conventions(tmp_path):
    torch.manual_seed(1)
    src = fr.get_network("mobilenetv3_small")
    bare = tmp_path / "w.pth"
    torch.save(src.state_dict(), bare)
    m = fr.load_model("mobilenetv3_small", str(bare), torch.device("cpu"))
    assert not m.training and ob.state_dict_sha256(m.state_dict()) == ob.state_dict_sha256(src.state_dict())
    import argparse
    ck = tmp_path / "w.ckpt"
    torch.save({"epoch": 3, "model": src.state_dict(), "args": argparse.Namespace(lr=0.1)}, ck)
    m2 = fr.load_model("mobilenetv3_small", str(ck), torch.device("cpu"))
    assert ob.state_dict_sha256(m2.state_dict()) == ob.state_dict_sha256(src.state_dict())
    # the oracle loads the same checkpoint (state_dict-identical)
    ob.build("mobilenetv3_small").load_state_dict(torch.load(bare))
    with pytest.raises(ValueError):
        fr.get_network("resnet50")
    with pytest.raises(RuntimeError):
        fr.load_model("mobilenetv3_small", str(tmp_path / "missing.pth"), torch.device("cpu"))
    with pytest.raises(RuntimeError):
        fr.load_model("mobilenetv2", str(bare), torch.device("cpu"))          # wrong architecture: strict load fails
    with pytest.raises(FileNotFoundError):
        fr.extract_features(m, torch.device("cpu"), str(tmp_path / "nope.jpg"))


def test_no_cpu_fallback():
    m = fr.get_network("mobilenetv1").eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 3, 112, 112))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        fr.cosine_pairs(torch.zeros(2, 512), torch.zeros(2, 512))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        fr.search.Matcher(torch.zeros(4, 512, dtype=torch.bfloat16))
    src = "".join(open(os.path.join(ROOT, "face-recognition_amd", f)).read()
                  for f in os.listdir(os.path.join(ROOT, "face-recognition_amd")) if f.endswith(".py"))
    assert "import oracle" not in src and "from oracle" not in src


def test_eval_missing_annotation_returns_empty_triple(tmp_path):
    m = fr.get_network("mobilenetv1")
    mean, preds, metrics = fr.evaluate.eval(m, device=torch.device("cpu"), val_root=str(tmp_path))
    assert mean == 0.0 and preds.size == 0 and metrics == {}


def test_transform_matches_oracle():
    from oracle import scoring as osc
    from PIL import Image
    a = np.random.default_rng(0).integers(0, 256, (112, 112, 3), dtype=np.uint8)
    assert torch.equal(fr.get_transform()(Image.fromarray(a)), osc.to_model_input(a))


def test_shard_ranges_partition_the_gallery():
    for G, P in ((100, 8), (10_000_001, 4), (7, 8)):
        r = [fr.search.shard_range(G, i, P) for i in range(P)]
        assert r[0][0] == 0 and r[-1][1] == G and all(a[1] == b[0] for a, b in zip(r, r[1:]))


def _gloo_worker(rank, world, port, q):
    try:
        _gloo_body(rank, world, port, q)
    except Exception as e:                      # surface the failure instead of a queue timeout
        q.put((rank, False, repr(e)))


def _gloo_body(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import search as osr
    G, Q, k = 1500, 40, 5
    g = osr.bf16_bits_to_f32(osr.gallery_rows(7, 11, 0, G, G, 4))
    qs = osr.bf16_bits_to_f32(osr.query_rows(11, 0, Q))
    lo, hi = fr.search.shard_range(G, rank, world)
    s, i, _ = osr.topk_cosine(qs, g[lo:hi], k, idx_offset=lo)           # stand-in for the local HIP search
    gs, gi = fr.search.all_gather_topk(torch.from_numpy(s).float(), torch.from_numpy(i))
    ms, mi = osr.merge_topk(gs.numpy(), gi.numpy(), k)
    fs, fi, _ = osr.topk_cosine(qs, g, k)
    ok = bool(np.array_equal(mi, fi))
    # the pipelined form ShardedMatcher runs: packed messages, one async all_gather per chunk of 24 queries
    # (two chunks, the second ragged), chunk t merged while chunk t+1 is searched
    calls = []

    def local_search(qc, sv, iv):
        cs, ci, _ = osr.topk_cosine(qc.numpy(), g[lo:hi], k, idx_offset=lo)
        sv.copy_(torch.from_numpy(cs).float()); iv.copy_(torch.from_numpy(ci))
        calls.append(("search", len(qc)))

    def merge(recv, Qc, out_s, out_i):
        views = [fr.search.packed_views(recv[p], Qc, k) for p in range(world)]
        cs, ci = osr.merge_topk(np.stack([v[0].numpy() for v in views]), np.stack([v[1].numpy() for v in views]), k)
        out_s.copy_(torch.from_numpy(cs).float()); out_i.copy_(torch.from_numpy(ci))
        calls.append(("merge", Qc))

    ps, pi = fr.search.exchange_topk_chunks(torch.from_numpy(qs), k, 24, world, None, local_search, merge)
    ok = ok and bool(np.array_equal(pi.numpy(), fi)) and bool(np.array_equal(ps.numpy(), ms.astype(np.float32)))
    ok = ok and calls == [("search", 24), ("search", 16), ("merge", 24), ("merge", 16)]     # chunk 0 merges after chunk 1's search
    q.put((rank, ok, tuple(gs.shape)))
    dist.destroy_process_group()


def test_sharded_exchange_world_size_2_gloo():
    """The N>1 path on CPU: shard ranges + all_gather of per-shard top-k + deterministic merge."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = [q.get(timeout=120) for _ in procs]
    [p.join(60) for p in procs]
    assert all(ok for _, ok, _ in res) and all(shape == (2, 40, 5) for _, _, shape in res), res


def test_bench_gpus_2_starts_two_ranks():
    """`python bench.py --gpus 2` without a rank environment must start two ranks itself (fresh children through
    torch.distributed.run, 127.0.0.1 rendezvous) and shard the 100 M-row gallery of BASELINE.json configs[4] between
    them.  --rendezvous-only stops after the ranks have found each other (gloo here: no GPU in this container)."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                               # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks"] == [0, 1]
    assert d["search"] == {"gallery_rows": 100_000_000, "shards": [[0, 50_000_000], [50_000_000, 100_000_000]]}
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rendezvous-only"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=600)
    d1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][0])
    assert d1["n_gpus"] == 1 and d1["search"]["shards"] == [[0, 10_000_000]]


def test_cpu_matcher_baseline_agrees_with_the_checker():
    """bench.py times oracle/search.py:topk_cosine_fp32_chunked as the matcher's CPU baseline; it must compute the
    same search as the float64 checker (topk_cosine) wherever the scores are separated by more than fp32 noise."""
    from oracle import search as osr
    g = osr.bf16_bits_to_f32(osr.gallery_rows(7, 11, 0, 20000, 20000, 8)).astype(np.float64)
    q = osr.bf16_bits_to_f32(osr.query_rows(11, 0, 64)).astype(np.float64)
    g = (g / np.linalg.norm(g, axis=1, keepdims=True)).astype(np.float32)
    q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    s, i = osr.topk_cosine_fp32_chunked(torch.from_numpy(q), torch.from_numpy(g), 5, chunk=3000)
    os_, oi, gap = osr.topk_cosine(q, g, 5)
    assert np.abs(s.numpy() - os_).max() <= 1e-5
    seps = np.minimum(np.abs(np.diff(os_, axis=1)).min(axis=1), gap)
    clear = seps > 1e-5
    assert clear.sum() >= 32 and np.array_equal(i.numpy()[clear], oi[clear])
    assert [int(v) for v in i[:8, 0]] == [osr.plant_row(t, 20000, 8) for t in range(8)]


def test_custom_ops_registered_and_refuse_cpu():
    """The hot path is reached through torch.library ops (namespace fr_hip) that exist for the
    CUDA/ROCm dispatch key only: a CPU tensor must be refused, never silently computed."""
    import torch
    for name in ("embed", "cosine_pairs", "l2norm_bf16", "topk_search", "topk_merge"):
        assert hasattr(torch.ops.fr_hip, name)
    with pytest.raises(NotImplementedError):
        torch.ops.fr_hip.cosine_pairs(torch.zeros(2, 8), torch.zeros(2, 8), 0.0, 0.35)
    with pytest.raises(NotImplementedError):
        torch.ops.fr_hip.topk_search(torch.zeros(2, 512, dtype=torch.bfloat16), torch.zeros(4, 512, dtype=torch.bfloat16), 5, 0)
    y = torch.ops.fr_hip.embed(torch.zeros(3, 3, 112, 112, device="meta"), 0, 512, 1, 1)
    assert tuple(y.shape) == (3, 512) and y.dtype == torch.float32


def test_gallery_format_roundtrip_and_validation(tmp_path):
    """On-disk gallery: shard files + id table, ranges across shard boundaries,
    per-rank loading = the matcher's contiguous sharding, JSON export in the reference's format
    (inference.py:153-160), and refusal of damaged directories."""
    import json
    from face_recognition_amd import gallery as G
    torch.manual_seed(3)
    rows = torch.nn.functional.normalize(torch.randn(1000, 512), dim=1).to(torch.bfloat16)
    ids = [f"pessoa_{i}_é.jpg" if i % 7 == 0 else f"img{i:04d}.png" for i in range(1000)]
    d = str(tmp_path / "g")
    with G.GalleryWriter(d, 512, shard_rows=384) as w:
        w.add_normalized(ids[:100], rows[:100])
        w.add_normalized(ids[100:], rows[100:])           # crosses two shard boundaries in one call
        with pytest.raises(RuntimeError):
            w.add(ids[:1], torch.zeros(1, 512))            # raw embeddings are normalised on the GPU only
        with pytest.raises(ValueError):
            w.add_normalized(ids[:2], rows[:1])
    g = G.open_gallery(d)
    assert len(g) == 1000 and g.dim == 512 and g.ids == ids
    assert sorted(f for f in os.listdir(d) if f.endswith(".bf16")) == ["rows-00000.bf16", "rows-00001.bf16", "rows-00002.bf16"]
    assert os.path.getsize(os.path.join(d, "rows-00002.bf16")) == (1000 - 768) * 1024
    assert torch.equal(g.load_rows("cpu"), rows)
    assert torch.equal(g.load_rows("cpu", 380, 771), rows[380:771])
    assert g.load_rows("cpu", 5, 5).shape == (0, 512)
    with pytest.raises(IndexError):
        g.load_rows("cpu", 10, 1001)
    parts = [g.load_for_rank("cpu", r, 3) for r in range(3)]
    assert [p[1] for p in parts] == [fr.search.shard_range(1000, r, 3)[0] for r in range(3)]
    assert torch.equal(torch.cat([p[0] for p in parts]), rows)
    assert g.lookup(torch.tensor([[0, 999, -1]])) == [[ids[0], ids[999], None]]
    # the reference's JSON format
    j = str(tmp_path / "out" / "embeddings.json")
    G.export_json(g, j, chunk=300)
    back = json.load(open(j))
    assert list(back.keys()) == ids and back[ids[7]] == rows[7].to(torch.float32).tolist()
    # an unfinished or damaged gallery does not open
    with pytest.raises(FileExistsError):
        G.GalleryWriter(d)
    half = str(tmp_path / "half")
    try:
        with G.GalleryWriter(half, 512) as w:
            w.add_normalized(ids[:3], rows[:3])
            raise KeyboardInterrupt
    except KeyboardInterrupt:
        pass
    with pytest.raises(FileNotFoundError):
        G.open_gallery(half)
    with open(os.path.join(d, "rows-00001.bf16"), "ab") as f:
        f.write(b"\0\0")
    with pytest.raises(ValueError):
        G.open_gallery(d)


def test_alignment_matrix_and_oracle_warp_properties():
    """SURVEY.md 8(f) rank 4 (utils/face_utils.py:6-71).  cv2 / scikit-image are not importable here, so the oracle
    (oracle/align.py, parity unpinned) is held to what the algorithms guarantee by construction: the similarity fit
    recovers a known transform, the template maps to itself, and the fixed-point warp reproduces source pixels exactly
    wherever the sampling positions are integers (identity, integer shifts), with zeros outside the image.  The host
    layer's batched estimate_norm must agree with the oracle's per-face restatement."""
    from oracle import align as oa
    rng = np.random.default_rng(0)
    tmpl = oa.REFERENCE_ALIGNMENT.astype(np.float64)
    assert np.allclose(oa.estimate_norm(tmpl), [[1, 0, 0], [0, 1, 0]], atol=1e-6)
    n = 64
    s, th, t = rng.uniform(0.4, 3.0, n), rng.uniform(-np.pi, np.pi, n), rng.uniform(-50, 300, (n, 2))
    R = np.stack([np.stack([np.cos(th), -np.sin(th)], -1), np.stack([np.sin(th), np.cos(th)], -1)], 1)
    lm = s[:, None, None] * np.einsum("nij,kj->nki", R, tmpl) + t[:, None, :]          # template seen through a similarity
    M = fr.align.estimate_norm(lm)
    back = np.einsum("nij,nkj->nki", M[:, :, :2], lm) + M[:, None, :, 2]
    assert np.abs(back - tmpl[None]).max() < 1e-6                                       # exact data: exact recovery
    noisy = lm + rng.normal(0, 1.5, lm.shape)
    A = fr.align.estimate_norm(noisy)
    B = np.stack([oa.estimate_norm(l) for l in noisy])
    assert np.abs(A - B).max() < 1e-9
    assert np.abs(fr.align.estimate_norm(noisy[3]) - B[3]).max() < 1e-9                 # single-face form
    assert np.abs(fr.align.invert_affine(B) - np.stack([oa.invert_affine(b) for b in B])).max() == 0
    with pytest.raises(AssertionError):
        fr.align.estimate_norm(np.zeros((4, 2)))
    with pytest.raises(AssertionError):
        fr.align.estimate_norm(tmpl, image_size=100)
    assert np.allclose(fr.align.estimate_norm(tmpl * (224 / 112.0), 224), [[1, 0, 0], [0, 1, 0]], atol=1e-6)
    # the warp at integer sampling positions is a copy
    img = rng.integers(0, 256, size=(150, 200, 3), dtype=np.uint8)
    assert np.array_equal(oa.warp_affine_u8(img, np.array([[1, 0, 0], [0, 1, 0.0]])), img[:112, :112])
    sh = oa.warp_affine_u8(img, np.array([[1, 0, -30], [0, 1, -20.0]]))                 # dst(x,y) = src(x+30, y+20)
    assert np.array_equal(sh, img[20:132, 30:142])
    out = oa.warp_affine_u8(img, np.array([[1, 0, 50], [0, 1, 60.0]]))                  # dst(x,y) = src(x-50, y-60)
    assert not out[:60].any() and not out[:, :50].any() and np.array_equal(out[60:, 50:], img[:52, :62])
    half = oa.warp_affine_u8(img, np.array([[1, 0, -0.5], [0, 1, 0.0]]))                # half-pixel shift: rounded mean of neighbours
    want = (img[:112, :112].astype(np.int32) + img[:112, 1:113] + 1) >> 1
    assert np.array_equal(half, want.astype(np.uint8))
