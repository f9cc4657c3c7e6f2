"""SA2 (fp32, factored) A/B: MPX_SA2_PERSISTENT=1 (persistent one-wave-per-SIMD kernel) vs 0 (two-wave kernel).
Prints a hash of the pooled SA2 rows and of the policy output (the two must agree bit for bit) and the kernel time.
usage: sa2_ab.py [B] [steps]"""
import hashlib
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
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
torch.manual_seed(0)
mdl = MotionPolicyNetwork().to(dev).eval()
prob = make_problem_batch(B, seed=1000, device=dev, kinds=("tabletop", "cubby", "dresser"), M1=40, M2=16, scene_pool=1024,
                          device_clouds=True)
aux = {}
with torch.no_grad():
    dq = mdl(prob["xyz"], prob["q_norm"], aux=aux)
h = lambda t: hashlib.sha256(t.contiguous().cpu().numpy().tobytes()).hexdigest()[:16]
print(f"MPX_SA2_PERSISTENT={os.environ.get('MPX_SA2_PERSISTENT', '1')}: sa2 rows {h(aux['sa3_in'])}  dq {h(dq)}", flush=True)
eng = RolloutEngine(mdl, prob, rerender_scene=True, scene_seed=17)
eng.step()
torch.cuda.synchronize()
_lib.profile_start("mpx_sa_mlp_factored", "mpx_sa_mlp")
t0 = time.perf_counter()
for _ in range(steps):
    eng.step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / steps * 1e3
prof = _lib.profile_stop()
print(f"{B} envs: {ms:.2f} ms/step = {B / ms:.1f} k env-steps/s; SA2 {np.mean(prof['mpx_sa_mlp_factored']):.2f} ms, "
      f"SA1 {np.mean(prof['mpx_sa_mlp']):.2f} ms; q hash {h(eng.q)}")
