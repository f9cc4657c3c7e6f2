"""Phase times of one wave of sa2_bf16x3_persistent_kernel (s_memtime stamps of workgroup 100, wave 0): per 32-row tile
the cycles of layer 2 (96 MFMAs: matrix floor 3072) and of layer 3 (192 MFMAs: 6144) -- each interval includes one
stamp's own cost, printed separately -- first 19 tiles of the wave, at the
bench size.  usage: python tools/probes/sa2_bf16_phase_probe.py [B]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "motion-policy-networks_amd")]
import torch

from mpinets_amd import _lib
from mpinets_amd.model import MotionPolicyNetwork
from mpinets_amd.scenes import make_problem_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device("cuda:0")
torch.manual_seed(0)
mdl = MotionPolicyNetwork().to(dev).eval().set_precision("bf16x3")
prob = make_problem_batch(B, seed=1000, device=dev, kinds=("tabletop", "cubby", "dresser"), M1=40, M2=16, scene_pool=1024,
                          device_clouds=True)
with torch.no_grad():
    mdl(prob["xyz"], prob["q_norm"])
    torch.cuda.synchronize()
    probe = torch.zeros(128, dtype=torch.int64, device=dev)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    lib.mpx_sa2_bf16x3_set_probe.argtypes = [ctypes.c_void_p]
    lib.mpx_sa2_bf16x3_set_probe(probe.data_ptr())
    mdl(prob["xyz"], prob["q_norm"])
    torch.cuda.synchronize()
    lib.mpx_sa2_bf16x3_set_probe(None)
t = probe.cpu().numpy()
l2, l3, gap, own = [], [], [], []
for k in range(0, 76, 4):  # four stamps per tile: start, after layer 2, after layer 3, and one more straight after it
    if t[k + 3] == 0:
        break
    l2.append(int(t[k + 1] - t[k]))
    l3.append(int(t[k + 2] - t[k + 1]))
    own.append(int(t[k + 3] - t[k + 2]))
    if k + 4 < 80 and t[k + 4]:
        gap.append(int(t[k + 4] - t[k + 3]))
print("stamp cost (two stamps back to back):", own)
print("layer 2 ticks per tile:", l2)
print("layer 3 ticks per tile:", l3)
print("between tiles:", gap)
print("boundaries in the tile (0 / 1 / 2 = more):", [int(v) for v in t[96:96 + len(l2)]])
if l2:
    import numpy as np
    print("medians: layer 2 %d (floor 3072 cycles), layer 3 %d (floor 6144), between %d" % (np.median(l2), np.median(l3), np.median(gap or [0])))
