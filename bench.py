"""bench.py -- embeddings/sec of the HIP embed path (BASELINE.json configs[1]: mobilenetv3_large,
batch 1024, bf16, random-init weights, synthetic faces), one process per GPU.

    python bench.py --gpus N --steps K --warmup W

A "step" = one fr_embed pass over one resident batch (input fp32 NCHW already in HBM).  N > 1
(launched by torch.distributed.run) runs N independent replicas -- embedding extraction shards
trivially and has no collective -- and reports the aggregate (weak scaling).
The JSON line also carries:
  roofline      dominant kernel family, algorithmic bytes / measured per-layer HIP-event time
  whole_path    the same for the whole network (17.70 MB/img)
  cpu_baseline  the CPU oracle (PyTorch fp32 restatement of the reference) on this box's host cores
  search        1:N top-5 queries/sec, gallery sharded over the N ranks, RCCL all-gather merge
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s HBM3E peak
MFMA_PEAK_TF = 2500.0          # dense bf16
ALG_MB_PER_IMG = {"mobilenetv3_large": 17.70, "mobilenetv3_small": 5.74, "mobilenetv2": 26.6,
                  "mobilenetv1": 5.12, "sphere64": 19.8}     # SURVEY.md 8(d), 2 B/elem


def synthetic_batch(batch, device, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.rand((batch, 3, 112, 112), generator=g) * 2 - 1).to(device)


def layer_alg_bytes(plan, batch):
    """Algorithmic bytes of every plan layer: (input elems + output elems) * 2 B (bf16 figures of
    SURVEY.md 8(d); SE counts its two 1x1 convs only)."""
    out = []
    for i in range(plan.n_layers):
        c, h, w = plan.dims[i]
        src = plan.srcs[i]
        ci, hi, wi = (3, 112, 112) if src < 0 else plan.dims[src]
        if plan.ops[i] == 1:        # SE: fc1 (C->S) + fc2 (S->C) on 1x1
            d = plan.descs_cio[i]
            elems = 2 * (d[0] + d[1])
        else:
            elems = ci * hi * wi + c * h * w
        out.append(elems * 2 * batch)
    return out


def layer_alg_flops(plan, batch):
    """2 x MACs of every plan layer; SE = its two 1x1 convs."""
    out = []
    for i in range(plan.n_layers):
        c, h, w = plan.dims[i]
        cin, cout, k, groups = plan.descs_ck[i]
        if plan.ops[i] == 1:
            macs = 2 * cin * cout
        elif plan.ops[i] == 2:
            src = plan.srcs[i]
            ci, hi, wi = (3, 112, 112) if src < 0 else plan.dims[src]
            macs = ci * hi * wi * cout
        else:
            macs = cout * (cin // groups) * k * k * h * w
        out.append(2.0 * macs * batch)
    return out


def cpu_baseline(model_name, seconds=15.0):
    from oracle import backbones as ob
    net = ob.seeded(model_name, 0)
    x = synthetic_batch(64, "cpu")
    with torch.no_grad():
        net(x[:8])
        t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < seconds:
            net(x); n += 64
        dt = time.perf_counter() - t0
        t1 = time.perf_counter(); m = 0
        while time.perf_counter() - t1 < seconds / 3:
            net(x[m % 64:m % 64 + 1]); m += 1
        dt1 = time.perf_counter() - t1
    return {"value": round(n / dt, 1), "unit": "embeddings/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} images, batch 64, fp32 oracle (PyTorch CPU restatement of the reference), {dt:.1f} s",
            "batch1_value": round(m / dt1, 1), "host_cpus": os.cpu_count()}


def bench_search(dev, rank, world, steps, warmup, n_gallery, n_queries):
    import face_recognition_amd as fr
    lo, hi = fr.search.shard_range(n_gallery, rank, world)
    g = fr.search.synth_rows(7, lo, hi - lo, dev, plant_seed=11, n_plant=1024, n_gallery_total=n_gallery)
    q = fr.search.synth_rows(11, 0, n_queries, dev)
    m = fr.search.ShardedMatcher(g, n_gallery, rank, world, k=5)
    for _ in range(warmup):
        s, i = m.search(q)
    sync(dev, world)
    t0 = time.perf_counter()
    for _ in range(steps):
        s, i = m.search(q)
    sync(dev, world)
    dt = max_over_ranks(time.perf_counter() - t0, dev, world)
    from oracle import search as osr
    planted_ok = all(int(i[t, 0]) == osr.plant_row(t, n_gallery, 1024) for t in range(0, min(1024, n_queries), 97))
    qps = n_queries * steps / dt
    flops = 2.0 * n_gallery * 512 * qps
    return {"metric": "1:N top-5 queries/sec", "value": round(qps, 1), "unit": "queries/sec", "gallery_rows": n_gallery,
            "queries": n_queries, "dim": 512, "dtype": "bf16", "n_gpus": world, "ms_per_step": round(dt / steps * 1e3, 3),
            "planted_matches_found": bool(planted_ok),
            "roofline": {"bound": "mfma", "achieved": round(flops / 1e12, 1), "peak": MFMA_PEAK_TF * world,
                         "unit": "TFLOP/s", "frac": round(flops / 1e12 / (MFMA_PEAK_TF * world), 4)}}


def sync(dev, world):
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
    torch.cuda.synchronize(dev)


def max_over_ranks(v, dev, world):
    if not torch.distributed.is_initialized():
        return v
    t = torch.tensor([v], dtype=torch.float64, device=dev)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    return float(t.item())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--model", default="mobilenetv3_large")
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--input", default="f32", choices=["f32", "u8"],
                    help="f32: normalised fp32 NCHW tensors, the reference's model(tensor) contract (default); "
                         "u8: uint8 HWC faces through fr_embed_u8 (transform fused into the stem kernel)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-search", action="store_true")
    ap.add_argument("--gallery", type=int, default=0,
                    help="1:N gallery rows in total (default: 10M on one GPU = BASELINE.json configs[3]; 100M sharded over the ranks = configs[4])")
    ap.add_argument("--queries", type=int, default=65536)
    ap.add_argument("--layers", action="store_true", help="print the per-layer table to stderr")
    ap.add_argument("--no-layer-profile", action="store_true", help="skip the per-layer event profile (roofline omitted)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", 0)); world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    dist_on = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ        # under torchrun even a single rank goes through RCCL
    if dist_on:
        torch.distributed.init_process_group("nccl", device_id=dev)
    import face_recognition_amd as fr

    torch.manual_seed(0)
    model = fr.get_network(a.model).to(dev).eval()
    model.precision = a.precision
    x = synthetic_batch(a.batch, dev)
    ms_layer = rows = plan = None
    # fr_embed cuts a batch >= FR_SPLIT_MIN (128) into FR_SPLIT_PARTS (2) sub-batches on two streams: one kernel
    # launch processes one layer of ONE sub-batch, and that is what the per-launch figures below refer to
    split_min = int(os.environ.get("FR_SPLIT_MIN", "128"))
    parts = min(int(os.environ.get("FR_SPLIT_PARTS", "2")), 4)
    parts = parts if (split_min > 0 and parts > 1 and a.batch >= split_min) else 1
    pb = a.batch // parts
    if rank == 0 and not a.no_layer_profile:
        # ---- per-layer HIP-event timing on the launch stream (5 profiled passes, averaged); runs before the
        # warm-up, outside the timed region
        plan = model.plan(dev)
        g = fr.models._Graph(); model.lower(g)
        plan.descs_cio = [(int(d.cin), int(d.cout)) for d in g.descs]
        plan.descs_ck = [(int(d.cin), int(d.cout), int(d.kernel), max(1, int(d.groups))) for d in g.descs]
        reps = 5
        acc = None
        for _ in range(3):                       # untimed: clocks and caches warm before the averaged passes
            plan.profile(x[:pb])
        for _ in range(reps):
            rows = plan.profile(x[:pb])
            acc = [r[2] for r in rows] if acc is None else [p + r[2] for p, r in zip(acc, rows)]
        ms_layer = [v / reps for v in acc]
    if a.input == "u8":
        xin = ((x * 0.5 + 0.5) * 255).round().clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
        step = model.forward_u8
    else:
        xin, step = x, model
    with torch.no_grad():
        for _ in range(a.warmup):
            y = step(xin)
        sync(dev, world)
        t0 = time.perf_counter()
        for _ in range(a.steps):
            y = step(xin)
        sync(dev, world)
    dt = max_over_ranks(time.perf_counter() - t0, dev, world)
    ms = dt / a.steps * 1e3
    value = a.batch * a.steps * world / dt

    cfg_name = {("mobilenetv3_large", 1024): "BASELINE.json configs[1]", ("sphere64", 512): "BASELINE.json configs[2]"}.get(
        (a.model, a.batch), "not a BASELINE.json config")
    line = {"metric": "embeddings/sec", "value": round(value, 1), "unit": "embeddings/sec", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": a.precision, "data": "synthetic",
            "config": {"workload": f"{a.model} batch-{a.batch} {a.precision} embedding extraction, 112x112 synthetic faces"
                                   f"{' (uint8 HWC input, transform in the stem kernel)' if a.input == 'u8' else ''}, "
                                   f"random-init weights ({cfg_name})",
                       "global_batch": a.batch * world, "parallelism": f"replicas x{world} (no collective)"}}

    if rank == 0 and ms_layer is not None:
        alg = layer_alg_bytes(plan, pb)
        flp = layer_alg_flops(plan, pb)
        fam = {}
        carry_b = carry_f = 0.0
        for (name, kern, _, _), t, b, fl in zip(rows, ms_layer, alg, flp):
            if kern.startswith("(fused"):          # computed inside the next layer's kernel: its work counts there
                carry_b += b; carry_f += fl
                continue
            f = fam.setdefault(kern, [0.0, 0.0, 0, 0.0])
            f[0] += t; f[1] += b + carry_b; f[2] += 1; f[3] += fl + carry_f
            carry_b = carry_f = 0.0
        dom = max(fam, key=lambda k: fam[k][0])
        t_dom, b_dom, n_dom, f_dom = fam[dom]
        # arithmetic intensity of the dominant family against the machine balance decides the bound
        if f_dom / b_dom > MFMA_PEAK_TF * 1e12 / (HBM_PEAK_GBS * 1e9):
            ach = f_dom / (t_dom * 1e-3) / 1e12
            line["roofline"] = {"bound": "mfma", "kernel": dom, "launches_per_step": n_dom * parts, "images_per_launch": pb,
                                "avg_launch_ms": round(t_dom / n_dom, 4), "alg_flops_per_launch": int(f_dom / n_dom),
                                "achieved": round(ach, 1), "peak": MFMA_PEAK_TF, "unit": "TFLOP/s",
                                "frac": round(ach / MFMA_PEAK_TF, 4), "traffic": None}
        else:
            ach = b_dom / (t_dom * 1e-3) / 1e9
            line["roofline"] = {"bound": "hbm", "kernel": dom, "launches_per_step": n_dom * parts, "images_per_launch": pb,
                                "avg_launch_ms": round(t_dom / n_dom, 4), "alg_bytes_per_launch": int(b_dom / n_dom),
                                "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None}
        tf = os.path.join(ROOT, "profiles", f"hbm_traffic_{a.model}_b{a.batch}.json")
        if os.path.exists(tf):          # HBM bytes per launch from a separate rocprofv3 --pmc run (tools/hbm_traffic.py)
            k = json.load(open(tf)).get("kernels", {}).get(dom)
            if k:
                line["roofline"]["traffic"] = int(k["hbm_bytes_per_step"] / (n_dom * parts))
                line["roofline"]["traffic_source"] = os.path.relpath(tf, ROOT)
                # measured HBM rate of the same launches; for a kernel that fuses several layers the algorithmic
                # bytes above are those of the unfused layer graph, so `achieved` can exceed it
                line["roofline"]["traffic_GBps"] = round(line["roofline"]["traffic"] / (t_dom / n_dom * 1e-3) / 1e9, 1)
        if any(r[1].startswith("(fused") for r in rows) and dom == "dwp_kernel":
            line["roofline"]["note"] = ("dwp_kernel computes 2-3 layers per launch (expand 1x1 + depthwise + project 1x1); "
                                        "alg_bytes_per_launch counts every fused layer's input and output as SURVEY.md 8(d) "
                                        "defines them, the intermediate activations never reach HBM (see traffic)")
        if a.model in ALG_MB_PER_IMG:
            whole = ALG_MB_PER_IMG[a.model] * 1e6 * a.batch / (ms * 1e-3) / 1e9
            line["whole_path"] = {"bound": "hbm", "alg_MB_per_image": ALG_MB_PER_IMG[a.model], "achieved": round(whole, 1),
                                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(whole / HBM_PEAK_GBS, 4),
                                  "launches_per_step": plan.lib.fr_plan_num_launches(plan.handle) * parts,
                                  "sum_launch_ms": round(sum(ms_layer) * parts, 4)}
        line["kernel_families"] = {k: {"ms": round(v[0] * parts, 4), "GBps": round(v[1] / (v[0] * 1e-3) / 1e9, 1) if v[0] > 0 else 0,
                                        "TFLOPs": round(v[3] / (v[0] * 1e-3) / 1e12, 1) if v[0] > 0 else 0,
                                        "launches": v[2] * parts} for k, v in fam.items()}
        if a.layers:
            for (name, kern, _, _), t, b in zip(rows, ms_layer, alg):
                print(f"{name:36s} {kern:22s} {t:8.4f} ms {b / (t * 1e-3) / 1e9 if t > 0 else 0:9.1f} GB/s", file=sys.stderr)

    if not a.no_search:
        ng = a.gallery or (10_000_000 if world == 1 else 100_000_000)
        res = bench_search(dev, rank, world, max(1, min(a.steps, 2)), 1, ng, a.queries)
        if rank == 0:
            line["search"] = res
    if rank == 0 and not a.no_cpu_baseline and world == 1:
        line["cpu_baseline"] = cpu_baseline(a.model)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist_on:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
