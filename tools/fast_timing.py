"""Headline workload in the bf16x3 mode: ms/step and the grouped-MLP kernels' times (A/B builds via MPX_LIB_PATH).
usage: fast_timing.py [B] [steps] [noref]   (noref: skip the 64-environment fp32 / bf16x3 comparison -- counter passes)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "motion-policy-networks_amd")]
import numpy as np
import torch

from mpinets_amd import _lib
from mpinets_amd.model import MotionPolicyNetwork
from mpinets_amd.rollout import RolloutEngine
from mpinets_amd.scenes import make_problem_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
torch.manual_seed(0)
mdl = MotionPolicyNetwork().to(dev).eval()
prob = make_problem_batch(B, seed=1000, device=dev, kinds=("tabletop", "cubby", "dresser"), M1=40, M2=16, scene_pool=1024,
                          device_clouds=True)
if len(sys.argv) > 3 and sys.argv[3] == "noref":
    mdl.set_precision("bf16x3")
else:
    with torch.no_grad():
        ref = mdl(prob["xyz"][:64], prob["q_norm"][:64]).clone()
        mdl.set_precision("bf16x3")
        got = mdl(prob["xyz"][:64], prob["q_norm"][:64]).clone()
    print(f"lib {os.path.basename(_lib.LIB_PATH)}: |dq(bf16x3) - dq(fp32)| max = {(got - ref).abs().max().item():.2e}")
eng = RolloutEngine(mdl, prob, rerender_scene=True, scene_seed=17, resample_subset=True, subset_seed=23)
eng.step()
torch.cuda.synchronize()
names = ("mpx_sa_mlp_bf16x3", "mpx_sa_mlp_bf16x3_factored", "mpx_linear_bf16x3", "mpx_linear_bf16x3_to_pairs",
         "mpx_linear_bf16x3_pairs", "mpx_linear_rowmax_bf16x3_pairs", "mpx_sa3_front_bf16x3", "mpx_groupnorm_leaky_to_pairs", "mpx_linear", "mpx_fps",
         "mpx_ball_query", "mpx_sort_queries")
_lib.profile_start(*names)
t0 = time.perf_counter()
for _ in range(steps):
    eng.step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / steps * 1e3
prof = _lib.profile_stop()
print(f"{B} envs bf16x3: {ms:.2f} ms/step = {B / ms:.1f} k env-steps/s; " +
      ", ".join(f"{k[4:]} {np.sum(v) / steps:.2f}" for k, v in prof.items()))
