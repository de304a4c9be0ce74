"""Whole-network img/s of a backbone under the fusion switches (FR_FUSE, FR_FUSE3), per precision.
python tools/fuse_modes_probe.py <precision> [batch]"""
import os, sys, time, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 3:                       # child: model precision batch env...
    sys.path.insert(0, ROOT)
    import torch, face_recognition_amd as fr
    name, prec, batch = sys.argv[1], sys.argv[2], int(sys.argv[3])
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    m = fr.get_network(name).to(dev).eval()
    x = (torch.rand((batch, 3, 112, 112)) * 2 - 1).to(dev)
    with torch.no_grad():
        for _ in range(5): m(x, precision=prec)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): m(x, precision=prec)
        torch.cuda.synchronize()
    print(json.dumps({"img_s": round(batch * 20 / (time.perf_counter() - t0), 1)}))
    sys.exit(0)
prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
batch = sys.argv[2] if len(sys.argv) > 2 else "1024"
for name in ("mobilenetv1", "mobilenetv2", "mobilenetv3_small", "mobilenetv3_large"):
    row = []
    for label, env in (("unfused", {"FR_FUSE": "0"}), ("dw+pw", {"FR_FUSE3": "0"}), ("expand+dw+pw", {})):
        e = dict(os.environ); e.update(env)
        out = subprocess.run([sys.executable, __file__, name, prec, batch, "x"], env=e, capture_output=True, text=True, timeout=300)
        try:
            row.append(f"{label} {json.loads(out.stdout.strip().splitlines()[-1])['img_s']}")
        except Exception:
            row.append(f"{label} FAILED {out.stderr[-300:]}")
    print(f"{name:20s} {prec}  " + "   ".join(row), flush=True)
