"""Shard == single rank on the REAL engine (SURVEY.md section 8e, section 4 item iv).

Two ranks (gloo, both on the one GPU of the box) each run the headline step -- RolloutEngine(rerender_scene=True):
scene re-render + policy forward + joint update + FK cloud refresh + collision check -- over their half of one
global batch; the gathered joint angles, collision flags and slab rows must equal a single-rank run over the whole
batch BIT FOR BIT.  That holds because every random draw is keyed by the global environment id
(mpx_scene_cloud's env_offset) and no kernel couples environments.

Batch sizes: the dense layers pick their launch shape from the row count (split-K below 1025 rows and few output
tiles, csrc/dense.hip splitk_plan), which changes the fp32 summation order; E and 2E are chosen inside one regime
(24 | 48: both split the same way; 1040 | 2080: neither splits), like the 8192-per-rank bench shards.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("E,steps", [(24, 2), (1040, 2)])
def test_two_ranks_equal_one_rank_bit_for_bit(tmp_path, E, steps):
    sys.path.insert(0, HERE)
    from shard_worker import run_range

    out = str(tmp_path / "sharded.npz")
    env = dict(os.environ, MPX_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(HERE, "shard_worker.py"), str(E), str(steps), out]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    got = dict(np.load(out))
    ref = run_range(torch.device("cuda:0"), 0, 2 * E, 2 * E, steps)
    assert got["q"].shape == (2 * E, 7)
    for k, v in ref.items():
        np.testing.assert_array_equal(got[k], v.cpu().numpy(), err_msg=k)
    # the two halves are different problems (the comparison is not vacuous)
    assert not np.array_equal(got["q"][:E], got["q"][E:])


def test_scene_draw_is_keyed_by_global_env_id(oracle):
    """mpx_scene_cloud(env_offset=o) on rows [o, o+n) == rows [o, o+n) of the unsharded call == the oracle."""
    from mpinets_amd.scenes import make_scenes, sample_scene_clouds

    dev = torch.device("cuda:0")
    scn = make_scenes(12, 3, ("tabletop", "cubby", "dresser"), 40, 16)
    prims = {k: torch.from_numpy(v).to(dev) for k, v in scn.items()}
    full, a_full, l_full, _ = sample_scene_clouds(prims, 4096, 99, write_label=True, return_aux=True)
    for o, n in ((0, 5), (5, 7), (11, 1)):
        part = {k: v[o:o + n].contiguous() for k, v in prims.items()}
        got, a, lab, _ = sample_scene_clouds(part, 4096, 99, write_label=True, return_aux=True, env_offset=o)
        assert torch.equal(got, full[o:o + n]) and torch.equal(a, a_full[o:o + n]) and torch.equal(lab, l_full[o:o + n])
        pts, assign, labels, _ = oracle.scene_cloud({k: v[o:o + n] for k, v in scn.items()}, 4096, 99, env_offset=o)
        np.testing.assert_array_equal(a.cpu().numpy().view(np.uint16), assign)
        np.testing.assert_array_equal(lab.cpu().numpy(), labels)
        np.testing.assert_allclose(got[..., :3].cpu().numpy(), pts, rtol=0, atol=2e-6)


def test_problem_batch_rows_do_not_depend_on_the_shard():
    from mpinets_amd.scenes import make_problem_batch

    dev = torch.device("cuda:0")
    kw = dict(seed=8, device=dev, kinds=("tabletop", "cubby", "dresser"), M1=40, M2=16)
    for pool, clouds in ((None, False), (5, True), (None, True)):
        whole = make_problem_batch(14, scene_pool=pool, device_clouds=clouds, **kw)
        for o, n in ((0, 6), (6, 8)):
            part = make_problem_batch(n, scene_pool=pool, device_clouds=clouds, env_offset=o, total_envs=14, **kw)
            for k, v in whole.items():
                if torch.is_tensor(v) and v.size(0) == 14:
                    assert torch.equal(part[k], v[o:o + n]), (k, pool, clouds, o)


def test_pipelined_rollout_equals_single_engine():
    """rollout.PipelinedRollout (the batch in shares on their own HIP streams, a stage apart) leaves exactly the state one
    engine over the whole batch leaves -- shares of >= 1025 environments keep the whole batch's launch shapes."""
    from mpinets_amd.model import MotionPolicyNetwork
    from mpinets_amd.rollout import PipelinedRollout, RolloutEngine
    from mpinets_amd.scenes import make_problem_batch

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    mdl = MotionPolicyNetwork().to(dev).eval()
    B, steps = 2080, 3
    mk = lambda: make_problem_batch(B, seed=12, device=dev, kinds=("tabletop", "cubby", "dresser"), M1=40, M2=16, scene_pool=64,
                                    device_clouds=True, env_offset=7, total_envs=B + 7)
    one = RolloutEngine(mdl, mk(), rerender_scene=True, scene_seed=3)
    for _ in range(steps):
        one.step()
    prob = mk()
    pr = PipelinedRollout(mdl, prob, ways=2, rerender_scene=True, scene_seed=3)
    assert [e.env_offset for e in pr.engines] == [7, 7 + B // 2]
    pr.run(2)
    pr.run(steps - 2)  # (continues: the stagger is applied once, the step counters carry over)
    torch.cuda.synchronize()
    assert torch.equal(pr.q, one.q) and torch.equal(pr.q_norm, one.q_norm) and torch.equal(pr.flags, one.flags)
    assert torch.equal(prob["xyz"], one.xyz)  # the shares are views of the caller's slab: updated in place
