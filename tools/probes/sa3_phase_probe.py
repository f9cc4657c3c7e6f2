"""Phase times of one workgroup of mpx_sa3_chain (s_memtime stamps of workgroup 300, wave 0), 8192 environments.
Matrix-pipe floor per phase: layer 1 1088 MFMAs x 64 = 69.6 k cycles, layer 2 131 k, each half of layer 3 131 k."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "motion-policy-networks_amd")]
import torch

from mpinets_amd import _lib
from mpinets_amd.model import MotionPolicyNetwork

dev = torch.device("cuda:0")
torch.manual_seed(0)
mdl = MotionPolicyNetwork().to(dev).eval()
enc = mdl.point_cloud_encoder
B = 8192
x = torch.randn((B * 128, 272), device=dev)
pack = enc._sa3_pack(272)
out = torch.empty((B, 1024), device=dev)
probe = torch.zeros(32, dtype=torch.int64, device=dev)
lib = ctypes.CDLL(_lib.LIB_PATH)
lib.mpx_sa3_chain_probe.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                    ctypes.c_void_p, ctypes.c_void_p]
for _ in range(2):
    assert lib.mpx_sa3_chain_probe(x.data_ptr(), 272, B, pack.data_ptr(), out.data_ptr(), 1024, probe.data_ptr(), None) == 0
torch.cuda.synchronize()
t = probe.cpu().numpy()
names = ["stage X", "layer 1", "barrier + write-back", "layer 2", "barrier + write-back", "layer 3 half 0", "layer 3 half 1",
         "barrier"]
floor = [0, 69632, 0, 131072, 0, 131072, 131072, 0]
k = 0
for p in range(2):
    for n, f in zip(names, floor):
        d = int(t[k + 1] - t[k])
        print(f"pass {p} {n:22s} {d:8d} ticks" + (f"  (matrix floor {f}: {f / max(d, 1):.3f})" if f else ""))
        k += 1
print("total", int(t[k] - t[0]), "ticks; matrix floor", 2 * sum(floor), f"= {2 * sum(floor) / (t[k] - t[0]):.3f}")
