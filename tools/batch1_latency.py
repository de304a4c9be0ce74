import sys, os, time, torch
sys.path.insert(0, os.getcwd())
import face_recognition_amd as fr
dev = torch.device("cuda", 0)
torch.manual_seed(0)
for name in ("mobilenetv3_large", "sphere64"):
    m = fr.get_network(name).to(dev).eval()
    x1 = (torch.rand((1, 3, 112, 112)) * 2 - 1).to(dev)
    for prec in ("bf16", "f32"):
        with torch.no_grad():
            for _ in range(50): m(x1, precision=prec)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(300): m(x1, precision=prec)
            torch.cuda.synchronize()
        print(name, prec, "batch 1: %.4f ms" % ((time.perf_counter() - t0) / 300 * 1e3))
