"""Per-call time, shape and matrix-pipe fraction of every dense-layer launch of one policy forward (development aid)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "motion-policy-networks_amd")]
import torch

from mpinets_amd import _lib
from mpinets_amd.model import MotionPolicyNetwork
from mpinets_amd.scenes import make_problem_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device("cuda:0")
torch.manual_seed(0)
mdl = MotionPolicyNetwork().to(dev).eval()
prob = make_problem_batch(B, seed=1000, device=dev, kinds=("tabletop", "cubby", "dresser"), M1=40, M2=16, scene_pool=1024,
                          device_clouds=True)
calls = []
orig = _lib.call


def spy(name, *args):
    if name.startswith("mpx_linear"):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig(name, *args)
        e1.record()
        calls.append((name, args, e0, e1))
    else:
        orig(name, *args)


with torch.no_grad():
    mdl(prob["xyz"], prob["q_norm"])
    _lib.call = spy
    for mod in (sys.modules["mpinets_amd.model"], sys.modules["mpinets_amd.pointnet2"]):
        if hasattr(mod, "_lib"):
            mod._lib.call = spy
    mdl(prob["xyz"], prob["q_norm"])
torch.cuda.synchronize()
tot = 0.0
for name, a, e0, e1 in calls:
    ms = e0.elapsed_time(e1)
    tot += ms
    ints = [x for x in a if isinstance(x, int) and not isinstance(x, bool) and 0 < x < (1 << 31)]
    # mpx_linear(x, ldx, w, b, M, K, N, relu, out, ldo): the four sizes after the pointers
    print(f"{name:28s} {ms:8.3f} ms  ints {ints[:8]}")
print(f"total {tot:.3f} ms over {len(calls)} calls")
