import sys, torch, numpy as np
sys.path.insert(0, '.')
import face_recognition_amd as fr
from oracle import scoring as osc
dev = torch.device('cuda:0')
torch.manual_seed(0)
for name, batch in (("mobilenetv3_small", 131100), ("mobilenetv1", 70000), ("sphere20", 66000)):
    m = fr.get_network(name).to(dev).eval()
    rng = np.random.default_rng(3)
    u8 = torch.from_numpy(rng.integers(0, 256, size=(11, 112, 112, 3), dtype=np.uint8)).to(dev)
    reps = (batch + 10) // 11
    big = u8.repeat(reps, 1, 1, 1)[:batch].contiguous()
    with torch.no_grad():
        y = m.forward_u8(big, precision="bf16")
        y11 = m.forward_u8(u8, precision="bf16")
    torch.cuda.synchronize()
    want = y11.repeat(reps, 1)[:batch]
    bad = (y != want).any(dim=1).nonzero().flatten()
    print(name, batch, "mismatching rows:", bad.numel(), bad[:8].tolist(), "finite", bool(torch.isfinite(y).all()), flush=True)
    del y, big, want, m
    torch.cuda.empty_cache()
