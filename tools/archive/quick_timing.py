"""Per-stage timings of the hot path on one GPU (development aid; bench.py is the contract)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "motion-policy-networks_amd")]
import torch

from mpinets_amd import _lib
from mpinets_amd.geometry import TorchCuboids, TorchCylinders
from mpinets_amd.model import MotionPolicyNetwork
from mpinets_amd.robot import FrankaCollisionSampler, FrankaSampler
from mpinets_amd.scenes import linear_trajectories, make_problem_batch

dev = torch.device("cuda:0")


def timeit(fn, n=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    torch.manual_seed(0)
    t0 = time.time()
    prob = make_problem_batch(B, seed=0, device=dev, scene_pool=64)
    print(f"setup B={B}: {time.time()-t0:.1f}s")
    mdl = MotionPolicyNetwork().to(dev).eval()
    xyz, qn = prob["xyz"], prob["q_norm"]
    enc = mdl.point_cloud_encoder
    sa1, sa2, sa3 = enc.SA_modules
    lib = _lib
    N = xyz.size(1)
    idx1 = torch.empty((B, 512), dtype=torch.int32, device=dev)
    xyz1 = torch.empty((B, 512, 3), device=dev)
    nbr1 = torch.empty((B, 512, 128), dtype=torch.int32, device=dev)
    cnt1 = torch.empty((B, 512), dtype=torch.int32, device=dev)
    cnt2 = torch.empty((B, 128), dtype=torch.int32, device=dev)
    ord1 = torch.empty(B * 512, dtype=torch.int32, device=dev)
    ord2 = torch.empty(B * 128, dtype=torch.int32, device=dev)
    scr = torch.empty(128, dtype=torch.int32, device=dev)
    f1 = torch.empty((B, 512, 64), device=dev)
    w1 = sa1._packed.get(sa1.convs(), 1)
    w2 = sa2._packed.get(sa2.convs(), 64)
    idx2 = torch.empty((B, 128), dtype=torch.int32, device=dev)
    xyz2 = torch.empty((B, 128, 3), device=dev)
    nbr2 = torch.empty((B, 128, 128), dtype=torch.int32, device=dev)
    f2 = torch.empty((B, 128, 256), device=dev)
    r = {}
    r["fps1"] = timeit(lambda: lib.call("mpx_fps", lib.ptr(xyz), B, N, 4, 512, lib.ptr(idx1), lib.ptr(xyz1), 3))
    r["ball1"] = timeit(lambda: lib.call("mpx_ball_query", lib.ptr(xyz1), 3, lib.ptr(xyz), 4, B, N, 512, 0.05, 128, lib.ptr(nbr1), lib.ptr(cnt1)))
    r["sa1_mlp"] = timeit(lambda: lib.call("mpx_sa_mlp", lib.ptr(xyz), 4, lib.ptr(xyz1), 3, lib.ptr(xyz) + 12, 4, 1, lib.ptr(nbr1), None, B, N, 512, 128, lib.ptr(w1), 64, 64, 64, lib.ptr(f1), 64, 0))
    r["sa1_mlp_elide"] = timeit(lambda: lib.call("mpx_sa_mlp", lib.ptr(xyz), 4, lib.ptr(xyz1), 3, lib.ptr(xyz) + 12, 4, 1, lib.ptr(nbr1), lib.ptr(cnt1), B, N, 512, 128, lib.ptr(w1), 64, 64, 64, lib.ptr(f1), 64, 0))
    r["fps2"] = timeit(lambda: lib.call("mpx_fps", lib.ptr(xyz1), B, 512, 3, 128, lib.ptr(idx2), lib.ptr(xyz2), 3))
    r["ball2"] = timeit(lambda: lib.call("mpx_ball_query", lib.ptr(xyz2), 3, lib.ptr(xyz1), 3, B, 512, 128, 0.3, 128, lib.ptr(nbr2), lib.ptr(cnt2)))
    r["sa2_mlp"] = timeit(lambda: lib.call("mpx_sa_mlp", lib.ptr(xyz1), 3, lib.ptr(xyz2), 3, lib.ptr(f1), 64, 64, lib.ptr(nbr2), None, B, 512, 128, 128, lib.ptr(w2), 128, 128, 256, lib.ptr(f2), 256, 0))
    r["sa2_mlp_elide"] = timeit(lambda: lib.call("mpx_sa_mlp", lib.ptr(xyz1), 3, lib.ptr(xyz2), 3, lib.ptr(f1), 64, 64, lib.ptr(nbr2), lib.ptr(cnt2), B, 512, 128, 128, lib.ptr(w2), 128, 128, 256, lib.ptr(f2), 256, 0))
    wb1 = sa1._packed.get(sa1.convs(), 1, "bf16x3")
    wb2 = sa2._packed.get(sa2.convs(), 64, "bf16x3")
    r["sa1_bf16x3"] = timeit(lambda: lib.call("mpx_sa_mlp_bf16x3", lib.ptr(xyz), 4, lib.ptr(xyz1), 3, lib.ptr(xyz) + 12, 4, 1, lib.ptr(nbr1), None, None, B, N, 512, 128, lib.ptr(wb1), 64, 64, 64, lib.ptr(f1), 64, 0))
    r["sort1"] = timeit(lambda: lib.call("mpx_sort_queries", lib.ptr(cnt1), B * 512, 128, lib.ptr(ord1), lib.ptr(scr)))
    r["sa1_bf16x3_elide"] = timeit(lambda: lib.call("mpx_sa_mlp_bf16x3", lib.ptr(xyz), 4, lib.ptr(xyz1), 3, lib.ptr(xyz) + 12, 4, 1, lib.ptr(nbr1), lib.ptr(cnt1), lib.ptr(ord1), B, N, 512, 128, lib.ptr(wb1), 64, 64, 64, lib.ptr(f1), 64, 0))
    r["sa2_bf16x3"] = timeit(lambda: lib.call("mpx_sa_mlp_bf16x3", lib.ptr(xyz1), 3, lib.ptr(xyz2), 3, lib.ptr(f1), 64, 64, lib.ptr(nbr2), None, None, B, 512, 128, 128, lib.ptr(wb2), 128, 128, 256, lib.ptr(f2), 256, 0))
    lib.call("mpx_sort_queries", lib.ptr(cnt2), B * 128, 128, lib.ptr(ord2), lib.ptr(scr))
    r["sa2_bf16x3_elide"] = timeit(lambda: lib.call("mpx_sa_mlp_bf16x3", lib.ptr(xyz1), 3, lib.ptr(xyz2), 3, lib.ptr(f1), 64, 64, lib.ptr(nbr2), lib.ptr(cnt2), lib.ptr(ord2), B, 512, 128, 128, lib.ptr(wb2), 128, 128, 256, lib.ptr(f2), 256, 0))
    with torch.no_grad():
        r["forward"] = timeit(lambda: mdl(xyz, qn), n=3, warm=1)
        mdl.set_precision("bf16x3")
        r["forward_bf16x3"] = timeit(lambda: mdl(xyz, qn), n=3, warm=1)
        mdl.set_precision("fp32")
    smp = FrankaSampler(dev)
    sub = smp.draw_subset(2048)
    r["fk_cloud"] = timeit(lambda: smp.sample_into(prob["q"], xyz, sub))
    cs = FrankaCollisionSampler(dev)
    cub = TorchCuboids(prob["cuboid_centers"], prob["cuboid_dims"], prob["cuboid_quats"])
    cyl = TorchCylinders(prob["cylinder_centers"], prob["cylinder_radii"], prob["cylinder_heights"], prob["cylinder_quats"])
    traj = torch.from_numpy(linear_trajectories(B, 50, 0)).to(dev)
    r["collision_T50"] = timeit(lambda: cs.check(traj, cub, cyl))
    r["collision_T1"] = timeit(lambda: cs.check(prob["q"], cub, cyl))
    flops = {"sa1_mlp": 1.107e9, "sa2_mlp": 1.892e9, "forward": 3.27e9, "sa1_bf16x3": 1.107e9, "sa2_bf16x3": 1.892e9,
             "forward_bf16x3": 3.27e9}
    for k, v in r.items():
        extra = f"  {flops[k]*B/v/1e9:8.1f} TFLOP/s" if k in flops else ""
        print(f"{k:14s} {v:10.3f} ms  ({v/B*1e3:8.2f} us/env){extra}")
    print(f"env-steps/s (forward only): {B/r['forward']*1e3:.0f}")


if __name__ == "__main__":
    main()
