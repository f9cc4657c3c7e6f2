"""Soak of the "exact by proof" sampling kernels.

The Morton-culled FPS skips chunks whose box-to-pick bound says they cannot change, the bucketed ball query looks at 9
grid columns instead of the cloud: both claim the plain kernels' indices BIT FOR BIT, and a bound that is off by one ulp
would only show on a few clouds in a thousand (the contraction-order A/B of tests/test_oracle_pointnet.py moves 1-2 of
1000 FPS sequences).  So: every one of the 8192 bench environments, at 5 consecutive steps of the closed loop with the
scene re-rendered each step (40 960 distinct clouds), both FPS stages and both ball queries, fast kernels
(mpx_set_variant(...,1), the default) against the plain kernels (variant 0) in ONE process on the SAME device buffers;
plus 256 of those clouds against the scalar oracle.  A hash log goes to gpurun_out/soak_hashes.json (copied to
profiles/ with the round's evidence).
"""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _h(t: torch.Tensor) -> str:
    return hashlib.sha256(t.contiguous().cpu().numpy().tobytes()).hexdigest()[:16]


def test_variant_switch_is_validated():
    from mpinets_amd import _lib

    lib = _lib.load()
    assert lib.mpx_get_variant(0) == 1 and lib.mpx_get_variant(1) == 1 and lib.mpx_get_variant(7) == -1
    assert lib.mpx_set_variant(7, 0) != 0 and lib.mpx_set_variant(0, 2) != 0
    assert lib.mpx_get_variant(0) == 1


@pytest.mark.parametrize("B,steps,n_oracle", [(8192, 5, 256)])
def test_fast_sampling_kernels_equal_plain_kernels_on_every_bench_cloud(oracle, B, steps, n_oracle):
    from concurrent.futures import ThreadPoolExecutor

    from mpinets_amd import _lib
    from mpinets_amd.model import MotionPolicyNetwork
    from mpinets_amd.pointnet2 import ball_query, furthest_point_sample
    from mpinets_amd.rollout import RolloutEngine
    from mpinets_amd.scenes import make_problem_batch

    lib = _lib.load()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    mdl = MotionPolicyNetwork().to(dev).eval()
    prob = make_problem_batch(B, seed=1000, device=dev, kinds=("tabletop", "cubby", "dresser"), M1=40, M2=16, scene_pool=1024,
                              device_clouds=True)  # (bench.py's batch)
    eng = RolloutEngine(mdl, prob, rerender_scene=True, scene_seed=17)

    def sample(variant):
        assert lib.mpx_set_variant(0, variant) == 0 and lib.mpx_set_variant(1, variant) == 0
        try:
            i1, x1 = furthest_point_sample(eng.xyz, 512, return_xyz=True)
            b1, c1 = ball_query(0.05, 128, eng.xyz, x1, return_counts=True)
            i2, x2 = furthest_point_sample(x1, 128, return_xyz=True)
            b2, c2 = ball_query(0.3, 128, x1, x2, return_counts=True)
            return dict(fps_idx1=i1, ball_idx1=b1, ball_cnt1=c1, fps_idx2=i2, ball_idx2=b2, ball_cnt2=c2)
        finally:
            lib.mpx_set_variant(0, 1), lib.mpx_set_variant(1, 1)

    log = {"envs": B, "steps": steps, "clouds": B * steps, "oracle_clouds": n_oracle, "hashes": []}
    pick = np.random.default_rng(3).permutation(B)[:n_oracle]  # the clouds that also go to the scalar oracle
    per_step = [pick[s::steps] for s in range(steps)]
    for s in range(steps):
        eng.step()  # re-render + policy forward + joint update + FK cloud refresh: a new cloud for every environment
        fast, plain = sample(1), sample(0)
        for k in fast:
            assert torch.equal(fast[k], plain[k]), f"step {s}: {k} differs between the fast and the plain kernel"
        log["hashes"].append({k: _h(v) for k, v in fast.items()})
        # a slice of this step's clouds against the scalar oracle
        rows = torch.from_numpy(per_step[s]).to(dev)
        x = eng.xyz[rows].cpu().numpy()
        chunks = np.array_split(np.arange(len(rows)), 16)
        with ThreadPoolExecutor(16) as ex:  # (ctypes drops the GIL)
            o1 = np.concatenate(list(ex.map(lambda c: oracle.fps(x[c], 512), chunks)))
        np.testing.assert_array_equal(fast["fps_idx1"][rows].cpu().numpy(), o1)
        ctr = oracle.gather_points(x, o1)
        ob1, oc1 = oracle.ball_query(ctr, x, 0.05, 128, return_counts=True)
        np.testing.assert_array_equal(fast["ball_idx1"][rows].cpu().numpy(), ob1)
        np.testing.assert_array_equal(fast["ball_cnt1"][rows].cpu().numpy(), oc1)
        o2 = oracle.fps(ctr, 128)
        np.testing.assert_array_equal(fast["fps_idx2"][rows].cpu().numpy(), o2)
        ob2, oc2 = oracle.ball_query(oracle.gather_points(ctr, o2), ctr, 0.3, 128, return_counts=True)
        np.testing.assert_array_equal(fast["ball_idx2"][rows].cpu().numpy(), ob2)
        np.testing.assert_array_equal(fast["ball_cnt2"][rows].cpu().numpy(), oc2)
        del fast, plain
    # the clouds did change from step to step (the soak is not five times the same comparison)
    assert len({h["fps_idx1"] for h in log["hashes"]}) == steps
    log["result"] = "fast == plain on every index tensor of every cloud; oracle slice equal"
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "soak_hashes.json"), "w") as f:
        json.dump(log, f, indent=1)
