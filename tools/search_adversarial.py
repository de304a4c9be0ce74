"""How data-dependent is the 1:N search rate?  The per-lane top-k selection of search_rq_kernel skips its insertion
code unless some lane of the wave beats its current k-th score (k_search.hip), so the rate depends on how often that
happens.  This measures it on three galleries of the same size:
  random      the benchmark's synthetic rows (insertions become rare after the first few tiles)
  ascending   every query's score grows with the row index: EVERY tile beats the current top-k of every lane
  descending  the best rows come first: no insertion after the first tiles (best case)
Round 3: the resident-query kernels walk the tiles of a split in a strided group order (k_search.hip: rq_tile_stride), so
the three orderings now run at the same rate; FR_SEARCH_WALK=0 restores the in-order walk for comparison.
Usage: python tools/search_adversarial.py [gallery_rows] [queries] [noise]   -> one JSON line"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import face_recognition_amd as fr  # noqa: E402


def ordered_gallery(G, dev, ascending, noise, chunk=1 << 20):
    """rows = cos(t_r) u + sin(t_r) w (+ a little noise) with the angle t_r monotone in r, so that the score of a
    query near u is cos(t_r): a steady trend over the whole gallery (0 -> 0.95, ~1e-7 per row).  The bf16 rounding of
    the rows adds ~1e-4 of score noise, so the running k-th best sits among the last ~1000 rows and a new 64-row tile
    beats it for some lane of nearly every wave: the insertion code runs on (nearly) every tile."""
    g = torch.Generator(device=dev).manual_seed(3)
    u = torch.randn(512, device=dev, generator=g); u = u / u.norm()
    w = torch.randn(512, device=dev, generator=g); w = w - (w @ u) * u; w = w / w.norm()
    out = torch.empty((G, 512), dtype=torch.bfloat16, device=dev)
    for lo in range(0, G, chunk):
        n = min(chunk, G - lo)
        r = torch.arange(lo, lo + n, device=dev, dtype=torch.float64) / G
        score = (r if ascending else 1 - r) * 0.95
        ct = score.to(torch.float32); st = torch.sqrt(1 - score * score).to(torch.float32)
        x = ct[:, None] * u[None, :] + st[:, None] * w[None, :]
        if noise > 0:
            x = x + torch.randn((n, 512), device=dev, generator=g) * (noise / 512 ** 0.5)
        out[lo:lo + n] = fr.search.normalize_rows_bf16(x)
    return out, u


def rate(m, q, steps=2):
    m.search(q); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        s, i = m.search(q)
    torch.cuda.synchronize()
    return q.shape[0] * steps / (time.perf_counter() - t0), s, i


def main():
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    Q = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    noise = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    dev = torch.device("cuda", 0)
    res = {"gallery_rows": G, "queries": Q, "k": 5, "noise": noise}
    g = fr.search.synth_rows(7, 0, G, dev, plant_seed=11, n_plant=1024, n_gallery_total=G)
    q = fr.search.synth_rows(11, 0, Q, dev)
    r, _, _ = rate(fr.search.Matcher(g, k=5), q)
    res["random"] = round(r, 1)
    del g
    for name, asc in (("ascending", True), ("descending", False)):
        g, u = ordered_gallery(G, dev, asc, noise)
        gen = torch.Generator(device=dev).manual_seed(5)
        qq = fr.search.normalize_rows_bf16(u[None, :] + 0.002 * torch.randn((Q, 512), device=dev, generator=gen))
        r, s, i = rate(fr.search.Matcher(g, k=5), qq)
        res[name] = round(r, 1)
        # the best rows sit at the end (ascending) / start (descending) of the gallery
        edge = (i >= G - G // 100).float().mean().item() if asc else (i < G // 100).float().mean().item()
        res[name + "_top5_in_expected_1pct"] = round(edge, 4)
        del g
    for k in ("random", "ascending", "descending"):
        res[k + "_TFLOPs"] = round(2.0 * G * 512 * res[k] / 1e12, 1)
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
