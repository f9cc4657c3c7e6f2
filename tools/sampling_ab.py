"""Sampling kernels A/B on the bench scenes: the culled FPS, the wave-per-query and the bucketed ball query (default)
against the plain kernels (MPX_FPS_CULL=0 MPX_BQ_WAVE=0 MPX_BQ_GRID=0) -- prints hashes of every index tensor of the
policy forward and of its output; the two settings must agree bit for bit.  usage: sampling_ab.py [B]"""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "motion-policy-networks_amd")]
import torch

from mpinets_amd.model import MotionPolicyNetwork
from mpinets_amd.scenes import make_problem_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda:0")
torch.manual_seed(0)
mdl = MotionPolicyNetwork().to(dev).eval()
prob = make_problem_batch(B, seed=1000, device=dev, kinds=("tabletop", "cubby", "dresser"), M1=40, M2=16, scene_pool=1024,
                          device_clouds=True)
aux = {}
with torch.no_grad():
    dq = mdl(prob["xyz"], prob["q_norm"], aux=aux)
h = lambda t: hashlib.sha256(t.contiguous().cpu().numpy().tobytes()).hexdigest()[:16]
sel = " ".join(f"{k}={os.environ.get(k, '1')}" for k in ("MPX_FPS_CULL", "MPX_BQ_WAVE", "MPX_BQ_GRID"))
print(sel + ": " + " ".join(f"{k} {h(aux[k])}" for k in ("fps_idx1", "ball_idx1", "ball_cnt1", "fps_idx2", "ball_idx2",
                                                          "ball_cnt2")) + f" dq {h(dq)}")
