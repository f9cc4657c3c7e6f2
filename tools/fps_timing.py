"""Farthest-point sampling alone (6272 -> 512 on bench-scene clouds) at several batch sizes, event-timed; prints a hash of
the indices so that A/B builds (MPX_LIB_PATH) can be compared for bit-identical output.
usage: fps_timing.py [B ...]   (default: 1 16 256 1024 8192)"""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "motion-policy-networks_amd")]
import torch

from mpinets_amd import _lib
from mpinets_amd.pointnet2 import furthest_point_sample
from mpinets_amd.scenes import make_problem_batch

sizes = [int(a) for a in sys.argv[1:]] or [1, 16, 256, 1024, 8192]
dev = torch.device("cuda:0")
prob = make_problem_batch(max(sizes), seed=1000, device=dev, kinds=("tabletop", "cubby", "dresser"), M1=40, M2=16,
                          scene_pool=1024, device_clouds=True)
for B in sizes:
    xyz = prob["xyz"][:B].contiguous()
    idx = furthest_point_sample(xyz, 512)
    torch.cuda.synchronize()
    reps = 20 if B <= 1024 else 5
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for r in range(reps):
        idx = furthest_point_sample(xyz, 512)
        ev[r + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[r].elapsed_time(ev[r + 1]) for r in range(reps))
    h = hashlib.sha256(idx.cpu().numpy().tobytes()).hexdigest()[:12]
    print(f"lib {os.path.basename(_lib.LIB_PATH)} fps B={B}: median {ts[len(ts) // 2]:.3f} ms  min {ts[0]:.3f}  idx sha {h}")
