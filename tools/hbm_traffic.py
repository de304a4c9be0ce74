"""Summarise two rocprofv3 counter runs (one --pmc FETCH_SIZE, one --pmc WRITE_SIZE, as the
micro-architecture guide prescribes: separate passes) of `bench.py --no-layer-profile --no-search
--no-cpu-baseline --steps S --warmup W` into HBM bytes per step and kernel.

    python tools/hbm_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <steps+warmup> <out.json> "<label>"

FETCH_SIZE is in KiB and is doubled: on gfx950 it tallies 128-byte requests at 64 bytes (confirmed
here with tools/micro/loadpat on a known byte count); WRITE_SIZE is exact for 16-byte-per-lane stores
(tools/micro/storepat)."""
import csv, json, re, sys, collections

def base(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"_Z\d+([A-Za-z_][A-Za-z_0-9]*?)I", name)
    return m.group(1) if m else re.split(r"[<(]", name)[0].strip()

def total(path, ctr):
    t = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == ctr:
            b = base(r["Kernel_Name"]); t[b] += float(r["Counter_Value"]); n[b] += 1
    return t, n

f, nf = total(sys.argv[1], "FETCH_SIZE")
w, nw = total(sys.argv[2], "WRITE_SIZE")
steps = int(sys.argv[3])
out = {}
for k in f:
    if k not in w or k.startswith("__amd"): continue
    fb, wb = f[k] * 1024 * 2 / steps, w[k] * 1024 / steps
    out[k] = {"launches_per_step": nf[k] / steps, "fetch_bytes_per_step": round(fb), "write_bytes_per_step": round(wb),
              "hbm_bytes_per_step": round(fb + wb)}
json.dump({"source": sys.argv[5] if len(sys.argv) > 5 else "", "steps_sampled": steps, "kernels": out}, open(sys.argv[4], "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_step"]):
    print(f"{k:24s} {v['launches_per_step']:6.1f} launches/step  fetch {v['fetch_bytes_per_step']/1e9:7.3f} GB  write {v['write_bytes_per_step']/1e9:7.3f} GB")
print("total GB/step", sum(v["hbm_bytes_per_step"] for v in out.values()) / 1e9)
