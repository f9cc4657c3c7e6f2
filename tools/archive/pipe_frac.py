import os, sys, time
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "motion-policy-networks_amd")]
import numpy as np, torch
from mpinets_amd import _lib
from mpinets_amd.model import MotionPolicyNetwork
from mpinets_amd.rollout import PipelinedRollout
from mpinets_amd.scenes import make_problem_batch
dev = torch.device("cuda:0"); torch.manual_seed(0)
mdl = MotionPolicyNetwork().to(dev).eval()
B = 8192
prob = make_problem_batch(B, seed=1000, device=dev, kinds=("tabletop", "cubby", "dresser"), M1=40, M2=16, scene_pool=1024, device_clouds=True)
pr = PipelinedRollout(mdl, prob, ways=2, rerender_scene=True, scene_seed=17)
pr.run(2); torch.cuda.synchronize()
_lib.profile_start("mpx_sa_mlp_factored", "mpx_sa_mlp", "mpx_fps", "mpx_ball_query")
t0 = time.perf_counter(); pr.run(5); torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
prof = _lib.profile_stop()
tiles = 0
for e in pr.engines:
    c = e.last_counts[1]
    rows4 = (c.clamp(1, 128) + 3) // 4 * 4
    tiles += int(((rows4.reshape(-1, 8).sum(1) + 31) // 32).sum().item())
sa2 = np.sum(prof["mpx_sa_mlp_factored"]) / 5
fl = tiles * 32 * (128 * 128 + 128 * 256) * 2
print(f"pipelined {ms:.2f} ms/step; SA2 launches sum {sa2:.2f} ms/step ({len(prof['mpx_sa_mlp_factored'])} launches), frac {fl / (sa2 * 1e-3) / 1e12 / 157.3:.3f}; fps {np.sum(prof['mpx_fps'])/5:.2f} bq {np.sum(prof['mpx_ball_query'])/5:.2f} sa1 {np.sum(prof['mpx_sa_mlp'])/5:.2f}")
