"""Batch-assembly throughput (row N2; development aid): samples/s of PointCloudInstanceDataset.get_batch."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "motion-policy-networks_amd")]
import numpy as np
import torch

from mpinets_amd.data import DatasetType, PointCloudInstanceDataset
from mpinets_amd.scenes import linear_trajectories, make_scenes

n, B = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 1024
scn = make_scenes(n, 0, ("tabletop", "cubby", "dresser"), 40, 16)
arr = {"cuboid_dims": scn["cuboid_dims"], "cuboid_centers": scn["cuboid_centers"], "cuboid_quaternions": scn["cuboid_quats"],
       "cylinder_radii": scn["cylinder_radii"], "cylinder_heights": scn["cylinder_heights"],
       "cylinder_centers": scn["cylinder_centers"], "cylinder_quaternions": scn["cylinder_quats"],
       "hybrid_solutions": linear_trajectories(n, 50, 1)}
ds = PointCloudInstanceDataset(arr, "hybrid_solutions", 2048, 4096, 128, DatasetType.TRAIN, 0.03, device="cuda:0")
idx = np.random.default_rng(0).integers(0, len(ds), B)
for _ in range(2):
    ds.get_batch(idx)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    ds.get_batch(idx)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print(f"batch {B}: {dt*1e3:.2f} ms = {B/dt:.0f} samples/s (slab {B*6272*16/2**20:.0f} MiB written: {B*6272*16/dt/1e9:.1f} GB/s)")
