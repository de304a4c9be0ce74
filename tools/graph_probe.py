"""Small-batch latency of model(x): eager launches vs hipGraph replay (fr_embed_graph_*).
Usage (GPU): python tools/graph_probe.py [model] [precision]   -> one JSON line"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import face_recognition_amd as fr

name = sys.argv[1] if len(sys.argv) > 1 else "mobilenetv3_large"
prec = sys.argv[2] if len(sys.argv) > 2 else "auto"
dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = fr.get_network(name).to(dev).eval()
m.precision = prec


def host_ms(fn, reps=200):
    """host time of the call alone (the stream is drained between calls)"""
    tot = 0.0
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(); tot += time.perf_counter() - t0
    torch.cuda.synchronize()
    return tot / reps * 1e3


def timed(fn, reps):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


res = {"model": name, "precision": m.plan(dev).precision, "ms_per_forward": {}}
for b in (1, 2, 8, 32, 64, 128):
    x = (torch.rand((b, 3, 112, 112), device=dev) * 2 - 1)
    os.environ["FR_GRAPH_MAX"] = "0"
    y0 = m(x)
    eager = timed(lambda: m(x), 300)
    eager_host = host_ms(lambda: m(x))
    os.environ["FR_GRAPH_MAX"] = "1024"
    y1 = m(x)
    fwd = timed(lambda: m(x), 300)
    g = m.plan(dev).graph(b)
    raw = timed(g.launch, 300)
    fwd_host = host_ms(lambda: m(x))
    res["ms_per_forward"][b] = {"eager": round(eager, 4), "graph_forward": round(fwd, 4), "graph_replay_only": round(raw, 4),
                                "host_ms_eager": round(eager_host, 4), "host_ms_graph_forward": round(fwd_host, 4), "kernel_nodes": g.nodes, "bit_identical": bool(torch.equal(y0, y1))}
print(json.dumps(res), flush=True)
