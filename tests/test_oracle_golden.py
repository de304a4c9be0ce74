"""CPU: the oracle (oracle/) against the golden vectors captured from the real reference."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import backbones as ob
from oracle import scoring as osc
from oracle import search as osr


def cos_rows(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return (a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))


@pytest.mark.parametrize("name", ob.MODEL_NAMES)
def test_seeded_weights_match_reference_hash(name, golden_dir):
    sha = json.load(open(os.path.join(golden_dir, "state_sha256.json")))
    assert ob.state_dict_sha256(ob.seeded(name, 0).state_dict()) == sha[name]


@pytest.mark.parametrize("name", ["mobilenetv1", "mobilenetv2", "mobilenetv3_small", "mobilenetv3_large", "sphere20"])
def test_oracle_embeddings_equal_reference(name, golden_dir, golden_x):
    g = np.load(os.path.join(golden_dir, f"emb_{name}.npz"))
    with torch.no_grad():
        y = ob.seeded(name, 0)(golden_x).numpy()
    # same torch build, same ops: bit-exact; tolerate 1e-6 relative for a different CPU
    assert np.allclose(y, g["emb"], rtol=1e-5, atol=1e-5 * np.abs(g["emb"]).max())
    assert cos_rows(y, g["emb64"]).min() > 0.999999


def load_bncal(net, path):
    g = np.load(path)
    sd = net.state_dict()
    for i, k in enumerate(g["keys"]):
        sd[str(k)] = torch.from_numpy(g[f"s{i}"])
    net.load_state_dict(sd)
    return g["emb"]


@pytest.mark.parametrize("name", ["mobilenetv1", "mobilenetv3_small"])
def test_oracle_bn_calibrated_embeddings(name, golden_dir, golden_x):
    net = ob.seeded(name, 0)
    ref = load_bncal(net, os.path.join(golden_dir, f"bncal_{name}.npz"))
    with torch.no_grad():
        y = net(golden_x).numpy()
    assert np.allclose(y, ref, rtol=1e-4, atol=1e-4)


def trained_like_state(name, golden_dir):
    """The seeded state_dict under oracle.backbones.trained_like(seed 2024) (+ the committed BN statistics calibrated
    on those weights, MobileNets) and the REFERENCE's embeddings on it (tests/golden/make_golden.py)."""
    g = np.load(os.path.join(golden_dir, f"trained_{name}.npz"))
    sd = ob.trained_like(ob.seeded(name, 0).state_dict(), seed=2024)
    if "keys" in g:
        for i, k in enumerate(g["keys"]):
            sd[str(k)] = torch.from_numpy(g[f"s{i}"])
    return sd, g["emb"]


@pytest.mark.parametrize("name", ["mobilenetv1", "mobilenetv3_small", "mobilenetv3_large", "sphere20", "sphere36"])
def test_oracle_trained_like_embeddings(name, golden_dir, golden_x):
    """The oracle on the trained-like weight set equals the reference on the same weights (fixture written by the
    reference itself)."""
    sd, ref = trained_like_state(name, golden_dir)
    net = ob.seeded(name, 0)
    net.load_state_dict(sd)
    with torch.no_grad():
        y = net(golden_x).numpy()
    assert np.allclose(y, ref, rtol=1e-4, atol=1e-4 * np.abs(ref).max())
    assert cos_rows(y, ref).min() > 0.9999999


def test_top_block_taps(golden_dir, golden_x):
    g = np.load(os.path.join(golden_dir, "emb_mobilenetv3_small.npz"))
    net = ob.seeded("mobilenetv3_small", 0)
    names = [str(n) for n in g["tap_names"]]
    _, taps = ob.forward_with_taps(net, golden_x, names)
    for n, summ in zip(names, g["taps"]):
        t = taps[n].double()
        assert abs(t.mean().item() - summ[0]) <= 1e-6 * max(abs(summ[0]), 1e-30) + 1e-12 * abs(summ[1])
        assert abs(t.pow(2).mean().sqrt().item() - summ[1]) <= 1e-5 * summ[1]


def test_pair_formulas_and_decisions(golden_dir):
    g = np.load(os.path.join(golden_dir, "pairs.npz"))
    a, b = g["a"], g["b"]
    s_np = np.array([osc.compute_similarity(x, y) for x, y in zip(a, b)], np.float32)
    s_lfw = np.array([osc.lfw_similarity(torch.from_numpy(x), torch.from_numpy(y)).item() for x, y in zip(a, b)])
    assert np.array_equal(s_np, g["sim_np"])
    assert np.array_equal(s_lfw, g["sim_lfw"])
    assert np.array_equal(osc.decide(s_lfw), g["dec_lfw"])
    assert np.array_equal(s_np > 0.35, g["dec_np"])
    # raw-init magnitudes: the +1e-5 formula collapses to ~0, the eps-free one does not
    assert np.all(np.abs(s_lfw[40:48]) < 1e-6) and np.all(s_np[40:44] > 0.999)


def test_lfw_line_parsing():
    assert osc.parse_lfw_line("r", "Abel 1 4\n") == (os.path.join("r", "Abel", "Abel_0001.jpg"),
                                                       os.path.join("r", "Abel", "Abel_0004.jpg"), "1")
    assert osc.parse_lfw_line("r", "A 2 B 10")[2] == "0"
    assert osc.parse_lfw_line("r", "garbage") is None and osc.parse_lfw_line("r", "") is None


def test_synthetic_rows_and_search_oracle():
    G, Q, npl = 2048, 32, 8
    g = osr.bf16_bits_to_f32(osr.gallery_rows(7, 11, 0, G, G, npl))
    q = osr.bf16_bits_to_f32(osr.query_rows(11, 0, Q))
    assert np.allclose(np.linalg.norm(g, axis=1), 1, atol=5e-3)
    # shard-invariance of the generator
    part = osr.bf16_bits_to_f32(osr.gallery_rows(7, 11, 1000, 300, G, npl))
    assert np.array_equal(part, g[1000:1300])
    s, i, gap = osr.topk_cosine(q, g, 5)
    for t in range(npl):
        assert i[t, 0] == osr.plant_row(t, G, npl) and s[t, 0] > 0.9
    # merge of shard results == global result
    parts = [osr.topk_cosine(q, g[lo:hi], 5, idx_offset=lo) for lo, hi in ((0, 700), (700, 1500), (1500, G))]
    ms, mi = osr.merge_topk(np.stack([p[0] for p in parts]), np.stack([p[1] for p in parts]), 5)
    assert np.array_equal(mi, i) and np.allclose(ms, s)
