"""How full are the ball-query neighbourhoods? (development aid)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "motion-policy-networks_amd")]
import numpy as np, torch
from mpinets_amd.model import MotionPolicyNetwork
from mpinets_amd.scenes import make_problem_batch
dev = torch.device("cuda:0")
torch.manual_seed(0)
mdl = MotionPolicyNetwork().to(dev).eval()
for kinds in (("tabletop",), ("cubby",), ("dresser",)):
    prob = make_problem_batch(256, seed=3, device=dev, kinds=kinds, M1=40, device_clouds=True)
    aux = {}
    with torch.no_grad():
        mdl(prob["xyz"], prob["q_norm"], aux=aux)
    for name in ("ball_idx1", "ball_idx2"):
        nbr = aux[name]
        cnt = (nbr != nbr[..., :1]).sum(-1) + 1   # unique slots (pads repeat the first hit)
        tiles = (cnt + 31) // 32
        c = cnt.float()
        print(f"{kinds[0]:9s} {name}: cnt mean {c.mean():6.1f} median {c.median():5.0f} p90 {c.quantile(0.9):5.0f} max {c.max():4.0f} | "
              f"tiles mean {tiles.float().mean():.3f} of 4  hist {[int((tiles==t).sum()) for t in (1,2,3,4)]}")
