import time, torch, sys
sys.path.insert(0, '.')
import face_recognition_amd as fr
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = fr.get_network('mobilenetv3_large').to(dev).eval(); m.precision = 'bf16'
x = torch.randn(1024, 3, 112, 112, device=dev)
with torch.no_grad():
    for _ in range(10): m(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): y = m(x)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print(f"enqueue {1e3*(t1-t0)/50:.3f} ms/step, total {1e3*(t2-t0)/50:.3f} ms/step")
