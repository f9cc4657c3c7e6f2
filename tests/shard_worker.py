"""One rank of the sharded real-engine run used by tests/test_gpu_shard.py (NOT a test module).

usage (under torch.distributed.run, gloo, every rank on cuda:0):
    shard_worker.py ENVS_PER_RANK STEPS OUT.npz
Rank r owns global environments [r*E, (r+1)*E) of ONE global batch (same seed on every rank), runs STEPS
closed-loop steps of RolloutEngine(rerender_scene=True) and rank 0 saves the gathered state.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "motion-policy-networks_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

SEED, SCENE_SEED, KINDS = 321, 5, ("tabletop", "cubby", "dresser")


def run_range(dev, env_offset, envs, total, steps):
    """-> dict of tensors after `steps` steps over global environments [env_offset, env_offset + envs)."""
    from mpinets_amd.model import MotionPolicyNetwork
    from mpinets_amd.rollout import RolloutEngine
    from mpinets_amd.scenes import make_problem_batch

    torch.manual_seed(0)
    model = MotionPolicyNetwork().to(dev).eval()
    prob = make_problem_batch(envs, seed=SEED, device=dev, kinds=KINDS, M1=40, M2=16, scene_pool=64, device_clouds=True,
                              env_offset=env_offset, total_envs=total)
    eng = RolloutEngine(model, prob, rerender_scene=True, scene_seed=SCENE_SEED)
    xyz0 = eng.xyz.clone()
    for _ in range(steps):
        eng.step()
    torch.cuda.synchronize()
    return {"q": eng.q, "q_norm": eng.q_norm, "flags": eng.flags, "xyz0": xyz0[:, ::97].contiguous(),
            "xyz": eng.xyz[:, ::97].contiguous()}


def main():
    from mpinets_amd import shard

    E, steps, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    dev = torch.device("cuda:0")  # both ranks share the one GPU of the box
    rank, ws, _ = shard.init(backend="gloo", device=dev)
    envs = shard.env_range(rank, ws, E)
    res = run_range(dev, envs.start, E, E * ws, steps)
    shard.barrier()
    gathered = {k: shard.gather_to_rank0(v) for k, v in res.items()}
    if rank == 0:
        np.savez(out, **{k: v.cpu().numpy() for k, v in gathered.items()})
    shard.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
