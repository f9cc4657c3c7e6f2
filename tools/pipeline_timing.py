"""One engine on one stream vs PipelinedRollout (shares of the batch on their own streams, a stage apart).
usage: pipeline_timing.py [B] [steps]   (development aid; results go to profiles/r02_other_measurements.md)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "motion-policy-networks_amd")]
import torch

from mpinets_amd.model import MotionPolicyNetwork
from mpinets_amd.rollout import PipelinedRollout, RolloutEngine
from mpinets_amd.scenes import make_problem_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
torch.manual_seed(0)
mdl = MotionPolicyNetwork().to(dev).eval()
kinds = ("tabletop", "cubby", "dresser")


def problem():
    return make_problem_batch(B, seed=1000, device=dev, kinds=kinds, M1=40, M2=16, scene_pool=1024, device_clouds=True)


def timed(run):
    run(2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


one = RolloutEngine(mdl, problem(), rerender_scene=True, scene_seed=17)
ms1 = timed(lambda n: [one.step() for _ in range(n)])
print(f"one stream, {B} envs: {ms1:.2f} ms/step = {B / ms1:.1f} k env-steps/s", flush=True)
ref_q, ref_f = one.q.clone(), one.flags.clone()
del one
for ways, stagger in ((2, True), (2, False), (3, True), (4, True)):
    pr = PipelinedRollout(mdl, problem(), ways=ways, stagger=stagger, rerender_scene=True, scene_seed=17)
    ms = timed(pr.run)
    same = torch.equal(pr.q, ref_q) and torch.equal(pr.flags, ref_f)
    print(f"{ways} shares on {ways} streams, stagger {stagger}: {ms:.2f} ms/step = {B / ms:.1f} k env-steps/s "
          f"({(ms1 / ms - 1) * 100:+.1f} %), state == single engine: {same}", flush=True)
    del pr
