"""Closed-loop parity over the horizons the BASELINE configs name, and for the step bench.py times.

* the headline step -- ``RolloutEngine(rerender_scene=True)``: scene re-render with the per-step seed schedule,
  policy forward, joint update, FK cloud refresh, collision check -- against an oracle loop (configs[4]);
* a 50-step closed-loop rollout (configs[2]; gen_data.py:77, model.py:128-183) in fp32 and ``bf16x3``:
  TEACHER-FORCED per-step parity (the oracle evaluates the engine's own state of every step: bit-exact FPS /
  ball-query indices, next joint state within 1e-5) plus a FREE-RUNNING comparison that reports how the two
  trajectories drift and the first step at which any index differs (written to gpurun_out/horizon_report.json);
* configs[2]'s batch itself: 256 problems x 50 steps in both precisions -- finite, deterministic, slab invariants.
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 1e-5  # BASELINE.json north_star: "policy deltas within 1e-5"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def dev():
    return torch.device("cuda:0")


def _prims(prob, sl=slice(None)):
    cub = tuple(prob[k][sl].cpu().numpy() for k in ("cuboid_centers", "cuboid_dims", "cuboid_quats"))
    cyl = tuple(prob[k][sl].cpu().numpy() for k in ("cylinder_centers", "cylinder_radii", "cylinder_heights",
                                                     "cylinder_quats"))
    return cub, cyl


class OracleLoop:
    """The oracle's restatement of one closed-loop step (model.py:170-181 + :293-314 for the new configuration)."""

    def __init__(self, orc, model, eng, prob):
        from mpinets_amd import franka_tables as ft

        self.orc, self.lim = orc, ft.JOINT_LIMITS_REAL
        self.sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
        self.tp, self.tl = eng.sampler.table_pts.cpu().numpy(), eng.sampler.table_link.cpu().numpy()
        self.subset = eng.subset.cpu().numpy()
        c = eng.collision
        self.sc, self.sr, self.sl = c.centers.cpu().numpy(), c.radii.cpu().numpy(), c.links.cpu().numpy()
        self.cub, self.cyl = _prims(prob)

    def step(self, xyz, qn):
        """xyz [B,N,4] (robot rows rewritten in place), qn [B,7] -> (qn', q', flags, aux)."""
        orc = self.orc
        dq, aux = orc.policy_forward(self.sd, xyz, qn)
        qn2 = np.clip(qn + dq, -1, 1).astype(np.float32)
        q = orc.unnormalize(qn2, self.lim)
        T = orc.franka_fk(q)
        xyz[:, :len(self.subset), :3] = orc.transform_table(T, self.tp, self.tl, self.subset)
        flags, _ = orc.collision_flags(orc.transform_table(T, self.sc, self.sl)[:, None], self.sr, self.cub, self.cyl)
        return qn2, q, flags, aux


def _same_indices(cap, aux):
    return all(np.array_equal(cap[a].cpu().numpy(), aux[b][c]) for a, b, c in (
        ("fps_idx1", "sa1", "fps_idx"), ("ball_idx1", "sa1", "ball_idx"), ("fps_idx2", "sa2", "fps_idx"),
        ("ball_idx2", "sa2", "ball_idx")))


def test_rerender_step_matches_oracle_loop(oracle):
    """What bench.py times: 3 steps of RolloutEngine(rerender_scene=True) on mixed scenes, a shard with a non-zero
    environment offset, vs the oracle loop with the same seed schedule (scene_seed + 7919 * step, global env ids)."""
    from mpinets_amd.model import MotionPolicyNetwork
    from mpinets_amd.rollout import RolloutEngine
    from mpinets_amd.scenes import make_problem_batch

    torch.manual_seed(3)
    model = MotionPolicyNetwork().to(dev()).eval()
    B, off, seed = 3, 5, 17
    prob = make_problem_batch(B, seed=11, device=dev(), kinds=("tabletop", "cubby", "dresser"), M1=40, M2=16,
                              device_clouds=True, env_offset=off, total_envs=off + B)
    eng = RolloutEngine(model, prob, rerender_scene=True, scene_seed=seed)
    assert eng.env_offset == off
    eng.capture = {}
    loop = OracleLoop(oracle, model, eng, prob)
    scn = {k: prob[k].cpu().numpy() for k in prob if k.startswith(("cuboid_", "cylinder_"))}
    x, qn = prob["xyz"].cpu().numpy().copy(), prob["q_norm"].cpu().numpy().copy()
    hit = np.zeros(B, bool)  # eng.flags ORs over the steps
    for step in range(3):
        eng.step()
        pts, assign, _, _ = oracle.scene_cloud(scn, 4096, seed + 7919 * step, env_offset=off)
        x[:, 2048:6144, :3] = pts
        qn, q, flags, aux = loop.step(x, qn)
        # scene rows: the draw (obstacle ids) bit-exact, coordinates to fp32 rounding of the surface map
        np.testing.assert_array_equal(eng._scene_scratch[0].cpu().numpy().view(np.uint16), assign)
        np.testing.assert_allclose(eng.xyz[:, 2048:6144, :3].cpu().numpy(), pts, rtol=0, atol=2e-6)
        assert _same_indices(eng.capture, aux), f"step {step}: FPS / ball-query indices differ"
        np.testing.assert_allclose(eng.q_norm.cpu().numpy(), qn, rtol=0, atol=TOL)
        np.testing.assert_allclose(eng.q.cpu().numpy(), q, rtol=0, atol=5e-5)
        np.testing.assert_allclose(eng.xyz[:, :2048, :3].cpu().numpy(), x[:, :2048, :3], rtol=0, atol=5e-5)
        hit |= flags
        np.testing.assert_array_equal(eng.flags.cpu().numpy() != 0, hit)
        assert torch.equal(eng.xyz[:, :, 3], prob["xyz"][:, :, 3])  # label column untouched
    # a different offset is a different draw
    eng2 = RolloutEngine(model, dict(prob, xyz=prob["xyz"].clone()), rerender_scene=True, scene_seed=seed, env_offset=0)
    eng2.step()
    assert not torch.equal(eng2._scene_scratch[0], eng._scene_scratch[0])


def test_per_step_robot_subset_matches_oracle_loop(oracle):
    """RolloutEngine(resample_subset=True): the robot cloud's point subset is redrawn at every step, one draw for the
    batch -- the reference's loop (FrankaSampler.sample's per-call np.random.choice; model.py:170-181,
    run_inference.py:188-189) -- on the device.  4 steps vs the oracle loop with the oracle's restated draw: subsets
    index-exact, FPS / ball-query indices bit-exact, joint state within 1e-5; the single-call C path (mpx_rollout with
    subset_table_size) leaves the identical state."""
    from mpinets_amd.model import MotionPolicyNetwork
    from mpinets_amd.rollout import RolloutEngine
    from mpinets_amd.scenes import make_problem_batch

    torch.manual_seed(4)
    model = MotionPolicyNetwork().to(dev()).eval()
    B, sseed = 3, 2024
    mk = lambda: make_problem_batch(B, seed=19, device=dev(), kinds=("tabletop", "cubby", "dresser"), M1=40, M2=16,
                                    device_clouds=True)
    prob = mk()
    eng = RolloutEngine(model, prob, resample_subset=True, subset_seed=sseed)
    frozen = RolloutEngine(model, mk())  # the default: one subset for the whole rollout
    eng.capture = {}
    loop = OracleLoop(oracle, model, eng, prob)
    x, qn = prob["xyz"].cpu().numpy().copy(), prob["q_norm"].cpu().numpy().copy()
    P = eng.sampler.table_pts.size(0)
    seen = []
    for step in range(4):
        eng.step(), frozen.step()
        loop.subset = oracle.draw_subset(P, 2048, sseed, step)
        np.testing.assert_array_equal(eng.subset.cpu().numpy(), loop.subset)
        seen.append(loop.subset.copy())
        qn, q, flags, aux = loop.step(x, qn)
        assert _same_indices(eng.capture, aux), f"step {step}: FPS / ball-query indices differ"
        np.testing.assert_allclose(eng.q_norm.cpu().numpy(), qn, rtol=0, atol=TOL)
        np.testing.assert_allclose(eng.xyz[:, :2048, :3].cpu().numpy(), x[:, :2048, :3], rtol=0, atol=5e-5)
    assert not any(np.array_equal(seen[0], s) for s in seen[1:])  # a new subset every step
    assert not torch.equal(eng.xyz[:, :2048], frozen.xyz[:, :2048])  # ... which the frozen default does not do
    # the same four steps through mpx_rollout (one C call)
    nat = RolloutEngine(model, mk(), resample_subset=True, subset_seed=sseed)
    nat.run_native(4)
    assert torch.equal(nat.q, eng.q) and torch.equal(nat.q_norm, eng.q_norm) and torch.equal(nat.xyz, eng.xyz)
    assert torch.equal(nat.subset, eng.subset)


@pytest.fixture(scope="module")
def horizon_report():
    rep = {}
    yield rep
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "horizon_report.json"), "w") as f:
        json.dump(rep, f, indent=1)


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_50_step_rollout_vs_oracle(oracle, horizon_report, precision):
    from mpinets_amd.model import MotionPolicyNetwork
    from mpinets_amd.rollout import RolloutEngine
    from mpinets_amd.scenes import make_problem_batch

    torch.manual_seed(0)
    model = MotionPolicyNetwork().to(dev()).eval().set_precision(precision)
    B, L = 3, 50
    prob = make_problem_batch(B, seed=21, device=dev(), kinds=("tabletop", "cubby", "dresser"), M1=40, M2=16)
    eng = RolloutEngine(model, prob)
    eng.capture = {}
    loop = OracleLoop(oracle, model, eng, prob)
    x_free, qn_free = prob["xyz"].cpu().numpy().copy(), prob["q_norm"].cpu().numpy().copy()
    forced_err, free_err, flips_forced, first_free_flip = [], [], [], None
    hit = np.zeros(B, bool)  # eng.flags ORs over the steps: "any waypoint in collision" (model.py:293-314)
    for step in range(L):
        x_t, qn_t = eng.xyz.cpu().numpy().copy(), eng.q_norm.cpu().numpy().copy()  # the engine's state before the step
        eng.step()
        got = eng.q_norm.cpu().numpy()
        # teacher-forced: the oracle on exactly this state
        qn_o, q_o, flags_o, aux = loop.step(x_t, qn_t)
        forced_err.append(float(np.abs(got - qn_o).max()))
        if not _same_indices(eng.capture, aux):
            flips_forced.append(step)
        hit |= flags_o
        np.testing.assert_array_equal(eng.flags.cpu().numpy() != 0, hit, err_msg=f"collision flags, step {step}")
        # free-running: the oracle on its own trajectory
        qn_free, _, _, aux_f = loop.step(x_free, qn_free)
        free_err.append(float(np.abs(got - qn_free).max()))
        if first_free_flip is None and not _same_indices(eng.capture, aux_f):
            first_free_flip = step
    horizon_report[precision] = {
        "envs": B, "steps": L, "teacher_forced_max_abs_err_per_step": forced_err,
        "teacher_forced_worst": max(forced_err), "teacher_forced_index_mismatch_steps": flips_forced,
        "free_running_max_abs_err_per_step": free_err, "free_running_first_index_difference_step": first_free_flip,
        "free_running_err_at_step": {str(k): free_err[k - 1] for k in (1, 5, 10, 25, 50)},
        "what": "normalised joint state after each step, engine vs oracle; teacher-forced = the oracle evaluates the "
                "engine's own state of that step; free-running = both follow their own trajectory from the same start"}
    print(precision, "teacher-forced worst %.2e, free-running after 50 steps %.2e, first free-running index difference: %s"
          % (max(forced_err), free_err[-1], first_free_flip))
    # the per-step claim of the north star, at every step of the horizon
    assert not flips_forced, f"FPS / ball-query indices differ from the oracle on the same state at steps {flips_forced}"
    assert max(forced_err) <= TOL, max(forced_err)
    # free-running drift: rounding differences (~1e-7 per step) are fed back through the loop; as long as no sampled
    # index has flipped the two trajectories stay within 1e-4 of each other
    upto = L if first_free_flip is None else first_free_flip
    assert all(e <= 1e-4 for e in free_err[:upto]), free_err[:upto]
    assert np.isfinite(free_err).all()


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_config3_batch_256_x_50_steps(oracle, precision):
    """BASELINE configs[2]: 256 problems, full policy forward, 50-step rollout (bf16 -> the 1e-5-compliant bf16x3)."""
    from mpinets_amd import franka_tables as ft
    from mpinets_amd.model import MotionPolicyNetwork
    from mpinets_amd.rollout import RolloutEngine
    from mpinets_amd.scenes import make_problem_batch

    torch.manual_seed(0)
    model = MotionPolicyNetwork().to(dev()).eval().set_precision(precision)
    B, L = 256, 50

    def run():
        prob = make_problem_batch(B, seed=33, device=dev(), kinds=("tabletop",), M1=16, M2=16, scene_pool=64,
                                  device_clouds=True)
        eng = RolloutEngine(model, prob)
        xyz0 = prob["xyz"].clone()
        traj = eng.rollout(L)
        torch.cuda.synchronize()
        return eng, xyz0, traj

    eng, xyz0, traj = run()
    assert traj.shape == (B, L + 1, 7) and torch.isfinite(traj).all() and torch.isfinite(eng.xyz).all()
    lim = torch.as_tensor(ft.JOINT_LIMITS_REAL, device=dev())
    assert (traj >= lim[:, 0] - 1e-5).all() and (traj <= lim[:, 1] + 1e-5).all()
    assert (eng.q_norm.abs() <= 1).all()
    # slab invariants: scene + target rows and the label column never change, robot rows = FK cloud of the final q
    assert torch.equal(eng.xyz[:, 2048:], xyz0[:, 2048:]) and torch.equal(eng.xyz[:, :, 3], xyz0[:, :, 3])
    ref = oracle.transform_table(oracle.franka_fk(eng.q[:4].cpu().numpy()), eng.sampler.table_pts.cpu().numpy(),
                                 eng.sampler.table_link.cpu().numpy(), eng.subset.cpu().numpy())
    np.testing.assert_allclose(eng.xyz[:4, :2048, :3].cpu().numpy(), ref, rtol=0, atol=1e-6)
    assert not torch.equal(traj[:, -1], traj[:, 0])  # the policy moved the arms
    # deterministic: the same rollout again is bit-identical
    eng2, _, traj2 = run()
    assert torch.equal(traj, traj2) and torch.equal(eng.flags, eng2.flags)
    # first step of a few problems against the oracle
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    dq, _ = oracle.policy_forward(sd, xyz0[:2].cpu().numpy(), make_qn(B, dev())[:2])
    q1 = oracle.unnormalize(np.clip(make_qn(B, dev())[:2] + dq, -1, 1).astype(np.float32), ft.JOINT_LIMITS_REAL)
    np.testing.assert_allclose(traj[:2, 1].cpu().numpy(), q1, rtol=0, atol=5e-5)


def make_qn(B, device):
    from mpinets_amd.scenes import make_problem_batch

    return make_problem_batch(B, seed=33, device=device, kinds=("tabletop",), M1=16, M2=16, scene_pool=64,
                              device_clouds=True)["q_norm"].cpu().numpy()
