"""FPS timing at the model's sizes, fast kernels (variant 1) and plain kernels (variant 0). usage: fps_timing.py [B] [npoint]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "motion-policy-networks_amd")]
import torch
from mpinets_amd import _lib
from mpinets_amd.scenes import make_problem_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
NP = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dev = torch.device("cuda:0")
prob = make_problem_batch(B, seed=0, device=dev, scene_pool=64, device_clouds=True)
xyz = prob["xyz"]
idx = torch.empty((B, NP), dtype=torch.int32, device=dev)
nx = torch.empty((B, NP, 3), device=dev)
def run():
    _lib.call("mpx_fps", _lib.ptr(xyz), B, 6272, 4, NP, _lib.ptr(idx), _lib.ptr(nx), 3)
for variant in (1, 0):
    _lib.load().mpx_set_variant(0, variant)
    for _ in range(2): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    print(f"variant {variant}: fps 6272->{NP}, B={B}: {e0.elapsed_time(e1)/5:.3f} ms  checksum {int(idx.sum())}")
_lib.load().mpx_set_variant(0, 1)
