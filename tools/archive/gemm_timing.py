"""mpx_linear microbench at the model's shapes (development aid). usage: gemm_timing.py [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "motion-policy-networks_amd")]
import torch
from mpinets_amd.pointnet2 import linear
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = torch.device("cuda:0")
def t(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
shapes = [("sa2_pre", B*512, 128, 68), ("sa3_l1", B*128, 512, 272), ("sa3_l2", B*128, 512, 512), ("sa3_l3", B*128, 1024, 512), ("fc1", B, 4096, 1024),
          ("fc2", B, 2048, 4096), ("fc3", B, 2048, 2048), ("dec1", B, 512, 2112), ("dec2", B, 256, 512), ("qenc", B, 128, 128)]
tot = 0
for name, M, N, K in shapes:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev)
    ms = t(lambda: linear(x, w, b, 1, out=y))
    tot += ms
    print(f"{name:8s} M={M:8d} N={N:5d} K={K:5d}: {ms:8.3f} ms  {2*M*N*K/ms/1e9:7.1f} TFLOP/s")
print("total", tot)
