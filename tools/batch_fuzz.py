"""Position invariance over odd batch sizes: image i must get the same bits whatever the batch it sits in (tile tails,
sub-batch split at 128, pooled squeeze-excite sums).  Usage (GPU): python tools/batch_fuzz.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import face_recognition_amd as fr
dev = torch.device("cuda", 0)
bad = 0
for name, precs in (("mobilenetv3_large", ("bf16", "f32")), ("mobilenetv2", ("bf16", "f32")), ("mobilenetv1", ("bf16",)), ("mobilenetv3_small", ("bf16", "f32", "f24")), ("sphere20", ("bf16", "f32")), ("sphere36", ("bf16",))):
    torch.manual_seed(1)
    m = fr.get_network(name).to(dev).eval()
    x = (torch.rand((1000, 3, 112, 112), device=dev) * 2 - 1)
    for prec in precs:
        ref = m(x, precision=prec)
        assert torch.isfinite(ref).all()
        for b in (1, 2, 3, 5, 7, 31, 33, 63, 65, 127, 128, 129, 255, 257, 511, 513, 999):
            y = m(x[:b], precision=prec)
            ok = torch.equal(y, ref[:b])
            if not ok:
                d = (y - ref[:b]).abs().max().item() / ref.abs().max().item()
                print(name, prec, b, "differs, rel", d); bad += d > 1e-5
    print(name, "done", flush=True)
print("BAD", bad)
