"""Aggregate a rocprofv3 kernel_stats / kernel_trace CSV by kernel base name (template instantiations
folded together): calls, total ms, average us.  Usage: summarize_rocprof.py <kernel_trace.csv> [out.csv]"""
import csv, re, sys, collections

def base(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"_Z\d+([A-Za-z_][A-Za-z_0-9]*?)I", name)
    if m:
        return m.group(1)
    return re.split(r"[<(]", name)[0].strip()

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    b = base(r["Kernel_Name"])
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a = agg.setdefault(b, [0, 0])
    a[0] += 1; a[1] += d
out = sorted(agg.items(), key=lambda kv: -kv[1][1])
w = csv.writer(open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout)
w.writerow(["kernel", "calls", "total_ms", "avg_us"])
for k, (n, t) in out:
    w.writerow([k, n, round(t / 1e6, 4), round(t / n / 1e3, 2)])
