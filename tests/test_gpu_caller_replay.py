"""Caller replay: the reference's inference driver, call for call, on the mirror package -- vs the oracle.

``replay_*`` below restate the CALL SEQUENCE of /root/reference/mpinets/run_inference.py
(``make_point_cloud_from_primitives`` :93-134, ``rollout_until_success`` :137-191, the driver's sampler / model
set-up :258-292) written against the names that file imports, which here come from ``mpinets_amd`` -- the "import
swap" of INTEGRATION.md section 1.  The oracle side computes the same quantities from the oracle's own kinematics
and network on the host, consuming ``random`` / ``np.random`` in the same order (host RNG draws are the reference's:
label shuffle + surface samples + final choice, one robot subset per ``sample`` call, one gripper subset), so the
two trajectories can be compared number for number.
"""
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# ---- the import swap (what run_inference.py:40-44 would import) ------------------------
from mpinets_amd.geometry import construct_mixed_point_cloud  # noqa: E402
from mpinets_amd.model import MotionPolicyNetwork  # noqa: E402
from mpinets_amd.primitives import Cuboid, Cylinder  # noqa: E402
from mpinets_amd.robot import FrankaRobot, FrankaSampler  # noqa: E402
from mpinets_amd.utils import normalize_franka_joints, unnormalize_franka_joints  # noqa: E402

NUM_ROBOT_POINTS, NUM_OBSTACLE_POINTS, NUM_TARGET_POINTS = 2048, 4096, 128  # run_inference.py:52-54
MAX_ROLLOUT_LENGTH = 6  # (150 in the reference; the loop body is what is replayed)


class Target:
    """The attributes the driver reads from its SE3 target: ``.matrix`` and ``._xyz`` (run_inference.py:66,181)."""

    def __init__(self, matrix):
        self.matrix = np.asarray(matrix, dtype=np.float64)
        self._xyz = self.matrix[:3, 3]


def _angle_deg(Ra, Rb):  # |angle| of Ra Rb^T: what the quaternion expression of run_inference.py:183-186 measures
    return float(np.degrees(np.arccos(np.clip((np.trace(Ra @ Rb.T) - 1) / 2, -1, 1))))


def replay_make_point_cloud_from_primitives(q0, target, obstacles, fk_sampler):
    """Call sequence of run_inference.py:93-134."""
    obstacle_points = construct_mixed_point_cloud(obstacles, NUM_OBSTACLE_POINTS)
    robot_points = fk_sampler.sample(q0, NUM_ROBOT_POINTS)
    target_points = fk_sampler.sample_end_effector(
        torch.as_tensor(target.matrix).type_as(robot_points).unsqueeze(0), num_points=NUM_TARGET_POINTS)
    xyz = torch.cat((torch.zeros(NUM_ROBOT_POINTS, 4), torch.ones(NUM_OBSTACLE_POINTS, 4),
                     2 * torch.ones(NUM_TARGET_POINTS, 4)), dim=0)
    xyz[:NUM_ROBOT_POINTS, :3] = robot_points.float()
    xyz[NUM_ROBOT_POINTS:NUM_ROBOT_POINTS + NUM_OBSTACLE_POINTS, :3] = torch.as_tensor(obstacle_points[:, :3]).float()
    xyz[NUM_ROBOT_POINTS + NUM_OBSTACLE_POINTS:, :3] = target_points.float()
    return xyz


def replay_rollout_until_success(mdl, q0, target, point_cloud, fk_sampler):
    """Call sequence of run_inference.py:137-191 (early stop: 1 cm and 15 degrees)."""
    q = torch.as_tensor(q0).unsqueeze(0).float().cuda()
    assert q.ndim == 2
    trajectory = [q]
    q_norm = normalize_franka_joints(q)
    assert isinstance(q_norm, torch.Tensor)

    def sampler(config):
        return fk_sampler.sample(config, NUM_ROBOT_POINTS)

    for _ in range(MAX_ROLLOUT_LENGTH):
        q_norm = torch.clamp(q_norm + mdl(point_cloud, q_norm), min=-1, max=1)
        qt = unnormalize_franka_joints(q_norm)
        assert isinstance(qt, torch.Tensor)
        trajectory.append(qt)
        eff_pose = FrankaRobot.fk(qt.squeeze().detach().cpu().numpy(), eff_frame="right_gripper")
        if np.linalg.norm(eff_pose._xyz - target._xyz) < 0.01 and _angle_deg(eff_pose.matrix[:3, :3], target.matrix[:3, :3]) < 15:
            break
        samples = sampler(qt).type_as(point_cloud)
        point_cloud[:, :samples.shape[1], :3] = samples
    return np.asarray([t.squeeze().detach().cpu().numpy() for t in trajectory])


def _obstacles():
    return [Cuboid(center=[0.6, 0.0, 0.1], dims=[0.5, 1.2, 0.05], quaternion=[1, 0, 0, 0]),
            Cuboid(center=[0.55, 0.3, 0.25], dims=[0.1, 0.12, 0.25], quaternion=[0.9238795, 0, 0, 0.3826834]),
            Cylinder(center=[0.5, -0.25, 0.22], radius=0.07, height=0.2, quaternion=[1, 0, 0, 0]),
            Cuboid(center=[-0.35, 0.0, -0.025], dims=[0.6, 0.6, 0.05], quaternion=[1, 0, 0, 0])]


def _oracle_replay(orc, sd, q0, target, obstacles, table_pts, table_link, eef_table, seed):
    """The same driver on the host: oracle FK / network, the reference's host RNG draws in the same order."""
    from mpinets_amd import franka_tables as ft

    random.seed(seed)
    np.random.seed(seed)
    obstacle_points = construct_mixed_point_cloud(obstacles, NUM_OBSTACLE_POINTS)  # host NumPy on both sides (pinned)
    sub = np.random.choice(len(table_pts), NUM_ROBOT_POINTS, replace=False).astype(np.int32)  # FrankaSampler.sample
    robot = orc.transform_table(orc.franka_fk(q0[None]), table_pts, table_link, sub)[0]
    esub = np.random.choice(len(eef_table), NUM_TARGET_POINTS, replace=False)  # sample_end_effector
    tm = target.matrix.astype(np.float32)
    tgt = eef_table[esub] @ tm[:3, :3].T + tm[:3, 3]
    xyz = np.zeros((1, NUM_ROBOT_POINTS + NUM_OBSTACLE_POINTS + NUM_TARGET_POINTS, 4), np.float32)
    xyz[0, NUM_ROBOT_POINTS:NUM_ROBOT_POINTS + NUM_OBSTACLE_POINTS, 3] = 1
    xyz[0, NUM_ROBOT_POINTS + NUM_OBSTACLE_POINTS:, 3] = 2
    xyz[0, :NUM_ROBOT_POINTS, :3] = robot
    xyz[0, NUM_ROBOT_POINTS:NUM_ROBOT_POINTS + NUM_OBSTACLE_POINTS, :3] = obstacle_points[:, :3].astype(np.float32)
    xyz[0, NUM_ROBOT_POINTS + NUM_OBSTACLE_POINTS:, :3] = tgt
    slab0 = xyz.copy()
    lim = ft.JOINT_LIMITS_REAL
    qn = orc.normalize(q0[None].astype(np.float32), lim)
    traj = [q0.astype(np.float32)]
    for _ in range(MAX_ROLLOUT_LENGTH):
        dq, _ = orc.policy_forward(sd, xyz, qn)
        qn = np.clip(qn + dq, -1, 1).astype(np.float32)
        qt = orc.unnormalize(qn, lim)
        traj.append(qt[0])
        T = orc.franka_fk(qt)
        eff = orc.frames_to_4x4(T[:, ft.LINK_ID["right_gripper"]])[0]
        if np.linalg.norm(eff[:3, 3] - target._xyz) < 0.01 and _angle_deg(eff[:3, :3], target.matrix[:3, :3]) < 15:
            break
        sub = np.random.choice(len(table_pts), NUM_ROBOT_POINTS, replace=False).astype(np.int32)
        xyz[0, :NUM_ROBOT_POINTS, :3] = orc.transform_table(T, table_pts, table_link, sub)[0]
    return slab0, np.asarray(traj), xyz


@pytest.mark.parametrize("reach_target", [False, True])
def test_inference_driver_replay_matches_oracle(oracle, reach_target):
    from mpinets_amd import franka_tables as ft

    seed = 1234
    torch.manual_seed(6)
    mdl = MotionPolicyNetwork().cuda()  # (a checkpoint in the reference: run_inference.py:262-263)
    mdl.eval()
    if reach_target:  # a policy that does not move: the start pose IS the target -> success after the first step
        with torch.no_grad():
            mdl.decoder[6].weight.zero_()
            mdl.decoder[6].bias.zero_()
    cpu_fk_sampler = FrankaSampler("cpu", use_cache=True)      # run_inference.py:264 -- host-facing handle
    gpu_fk_sampler = FrankaSampler("cuda:0", use_cache=True)   # run_inference.py:265
    q0 = np.array([0.1, -0.8, 0.1, -2.0, 0.1, 2.0, 0.7], np.float32)  # inside FrankaRealRobot's limits (the YAML's default_q is not)
    assert (q0 > ft.JOINT_LIMITS_REAL[:, 0]).all() and (q0 < ft.JOINT_LIMITS_REAL[:, 1]).all()
    q_goal = q0 if reach_target else q0 + np.array([0.4, 0.3, -0.3, 0.2, 0.1, 0.3, -0.2], np.float32)
    target = Target(FrankaRobot.fk(q_goal, eff_frame="right_gripper").matrix)
    obstacles = _obstacles()

    random.seed(seed)
    np.random.seed(seed)
    with torch.no_grad():
        point_cloud = replay_make_point_cloud_from_primitives(torch.as_tensor(q0).unsqueeze(0), target, obstacles,
                                                              cpu_fk_sampler)
        assert point_cloud.device.type == "cpu" and point_cloud.shape == (6272, 4)  # built on the host, like the driver
        slab = point_cloud.unsqueeze(0).cuda()
        slab0 = slab.clone()
        trajectory = replay_rollout_until_success(mdl, q0, target, slab, gpu_fk_sampler)

    sd = {k: v.detach().cpu().numpy() for k, v in mdl.state_dict().items()}
    o_slab0, o_traj, o_slab = _oracle_replay(
        oracle, sd, q0, target, obstacles, gpu_fk_sampler.table_pts.cpu().numpy(), gpu_fk_sampler.table_link.cpu().numpy(),
        gpu_fk_sampler.eef_table.cpu().numpy(), seed)
    # the slab the driver built: labels exact, scene rows exact (same host function, same RNG), robot / target rows
    # to FK rounding
    np.testing.assert_array_equal(slab0[0, :, 3].cpu().numpy(), o_slab0[0, :, 3])
    np.testing.assert_array_equal(slab0[0, 2048:6144, :3].cpu().numpy(), o_slab0[0, 2048:6144, :3])
    np.testing.assert_allclose(slab0[0, :, :3].cpu().numpy(), o_slab0[0, :, :3], rtol=0, atol=2e-6)
    # the trajectory: same length (same early-stop decision), same waypoints
    assert trajectory.shape == o_traj.shape == ((2, 7) if reach_target else (MAX_ROLLOUT_LENGTH + 1, 7))
    np.testing.assert_allclose(trajectory, o_traj, rtol=0, atol=5e-5)
    # and the slab the loop left behind (robot rows of the last re-sample, in place)
    np.testing.assert_allclose(slab[0, :, :3].cpu().numpy(), o_slab[0, :, :3], rtol=0, atol=5e-5)
    assert torch.equal(slab[:, 2048:], slab0[:, 2048:])
