"""search_rq16_kernel (16x16x32 MFMA) against search_rq_kernel (32x32x16): agreement and rate.
python tools/search_mfma_probe.py [gallery_rows] [queries]"""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import face_recognition_amd as fr  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
dev = torch.device("cuda", 0)
g = fr.search.synth_rows(7, 0, G, dev, plant_seed=11, n_plant=1024, n_gallery_total=G)
q = fr.search.synth_rows(11, 0, Q, dev)
m = fr.search.Matcher(g, k=5)
res = {}
out = {}
for mode in ("32", "16", "32", "16"):
    os.environ["FR_SEARCH_MFMA"] = mode
    m.search(q); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        s, i = m.search(q)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    res.setdefault(mode, []).append(round(Q / dt, 1))
    out[mode] = (s, i)
s32, i32 = out["32"]; s16, i16 = out["16"]
res["max_score_diff"] = float((s32 - s16).abs().max())
res["index_mismatch_frac"] = float((i32 != i16).float().mean())
gap = (s32[:, :-1] - s32[:, 1:]).min(dim=1).values
clear = gap > 1e-5
res["index_mismatch_where_gaps_clear"] = int((i32[clear] != i16[clear]).sum())
res["TFLOPs"] = {k: [round(2.0 * G * 512 * v / 1e12, 1) for v in vs] for k, vs in res.items() if k in ("16", "32")}
print(json.dumps(res))
