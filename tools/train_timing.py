"""Training-step timing (row N1; development aid): forward + losses + backward + Adam at batch size B."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "motion-policy-networks_amd")]
import torch

from mpinets_amd.model import TrainingMotionPolicyNetwork
from mpinets_amd.scenes import make_problem_batch
from mpinets_amd.training import train_step

NAMES = ("cuboid_centers", "cuboid_dims", "cuboid_quats", "cylinder_centers", "cylinder_radii", "cylinder_heights",
         "cylinder_quats")


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    prec = sys.argv[3] if len(sys.argv) > 3 else "fp32"  # "bf16x3": set_training_precision (split-bf16 training GEMMs)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    prob = make_problem_batch(B, seed=0, device=dev, kinds=("tabletop", "cubby", "dresser"), M1=40, M2=16,
                              scene_pool=256, device_clouds=True)
    sup = torch.clamp(prob["q_norm"] + 0.05 * torch.randn(B, 7, device=dev), -1, 1)
    batch = {"xyz": prob["xyz"], "configuration": prob["q_norm"], "supervision": sup, **{k: prob[k] for k in NAMES}}
    mdl = TrainingMotionPolicyNetwork(2048, 1.0, 1.0).to(dev).set_training_precision(prec)
    opt = mdl.configure_optimizers()
    for _ in range(2):
        train_step(mdl, opt, batch)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = train_step(mdl, opt, batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    c1, c2 = mdl.point_cloud_encoder.last_counts
    print(f"B={B} {prec}: {dt*1e3:.1f} ms/step, {B/dt:.0f} samples/s, loss {loss.item():.5f}, "
          f"peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB, rows SA1 {int(c1.clamp(min=1).sum())} "
          f"SA2 {int(c2.clamp(min=1).sum())}")


if __name__ == "__main__":
    main()
