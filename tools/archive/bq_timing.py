"""Ball query timing at the first module's shape (development aid). usage: bq_timing.py [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "motion-policy-networks_amd")]
import torch
from mpinets_amd import _lib
from mpinets_amd.scenes import make_problem_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device("cuda:0")
prob = make_problem_batch(B, seed=1000, device=dev, kinds=("tabletop", "cubby", "dresser"), M1=40, M2=16, scene_pool=1024, device_clouds=True)
xyz = prob["xyz"]
idx = torch.empty((B, 512), dtype=torch.int32, device=dev)
nx = torch.empty((B, 512, 3), device=dev)
_lib.call("mpx_fps", _lib.ptr(xyz), B, 6272, 4, 512, _lib.ptr(idx), _lib.ptr(nx), 3)
nbr = torch.empty((B, 512, 128), dtype=torch.int32, device=dev)
cnt = torch.empty((B, 512), dtype=torch.int32, device=dev)
def run():
    _lib.call("mpx_ball_query", _lib.ptr(nx), 3, _lib.ptr(xyz), 4, B, 6272, 512, 0.05, 128, _lib.ptr(nbr), _lib.ptr(cnt))
for _ in range(2): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): run()
e1.record(); torch.cuda.synchronize()
print(f"lib {os.path.basename(_lib.LIB_PATH)}: ball query 6272 x 512, r = 5 cm, B={B}: {e0.elapsed_time(e1)/5:.3f} ms  checksum {int(nbr.sum())} {int(cnt.sum())}")
