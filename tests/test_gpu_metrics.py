"""Row N3: batched trajectory metrics vs the oracle (and vs hand-checkable cases)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def test_trajectory_metrics_match_oracle(oracle):
    from mpinets_amd import franka_tables as ft
    from mpinets_amd.geometry import TorchCuboids, TorchCylinders
    from mpinets_amd.metrics import BatchedEvaluator
    from mpinets_amd.robot import franka_fk, frames_to_matrix
    from mpinets_amd.scenes import linear_trajectories, make_scenes, random_configurations

    B, Tn = 40, 150  # > 64 waypoints: several passes per trajectory
    traj = linear_trajectories(B, Tn, 3)
    traj[1, 10, 2] = 3.5  # outside the published limits
    lengths = np.random.default_rng(0).integers(1, Tn + 1, B).astype(np.int32)
    lengths[0], lengths[2] = Tn, 1
    goal = random_configurations(B, 11)
    goal[3] = traj[3, lengths[3] - 1]  # env 3 ends exactly on its target
    tt = torch.from_numpy(traj).to(dev())
    targets = frames_to_matrix(franka_fk(torch.from_numpy(goal).to(dev()))[:, ft.LINK_ID["right_gripper"]]).contiguous()
    scn = make_scenes(B, 4, ("tabletop",), 16, 16)
    cub = TorchCuboids(*(torch.from_numpy(scn[k]).to(dev()) for k in ("cuboid_centers", "cuboid_dims", "cuboid_quats")))
    cyl = TorchCylinders(*(torch.from_numpy(scn[k]).to(dev()) for k in ("cylinder_centers", "cylinder_radii", "cylinder_heights", "cylinder_quats")))
    ev = BatchedEvaluator(dev())
    got = ev.evaluate_trajectories(tt, targets, torch.from_numpy(lengths).to(dev()), cub, cyl)
    ref = oracle.trajectory_metrics(traj, lengths, targets.cpu().numpy(), ft.JOINT_LIMITS_PUBLISHED)
    for k, tol in (("position_error", 1e-3), ("orientation_error", 2e-2), ("eff_position_path_length", 1e-4),
                   ("eff_orientation_path_length", 0.2)):
        np.testing.assert_allclose(got[k].cpu().numpy(), ref[k], rtol=1e-4, atol=tol, err_msg=k)
    np.testing.assert_array_equal(got["joint_limit_violation"].cpu().numpy(), ref["joint_limit_violation"])
    np.testing.assert_array_equal(got["self_collision"].cpu().numpy(), ref["self_collision"])
    assert bool(got["joint_limit_violation"][1]) == (lengths[1] > 10)
    assert got["position_error"][3] < 1e-3 and got["orientation_error"][3] < 0.1
    assert got["eff_position_path_length"][2] == 0  # a single waypoint has no path
    # collision flag = the fused swept-sphere check on the valid part of each trajectory
    frozen = traj.copy()
    for b in range(B):
        frozen[b, lengths[b]:] = frozen[b, lengths[b] - 1]
    np.testing.assert_array_equal(got["collision"].cpu().numpy(),
                                  ev.collision_sampler.check(torch.from_numpy(frozen).to(dev()), cub, cyl).cpu().numpy())
    s = got["success"].cpu().numpy()
    assert s.dtype == bool and (~s | ~got["physical_violations"].cpu().numpy()).all()


def test_neutral_pose_is_clean_and_folded_arm_self_collides(oracle):
    from mpinets_amd import franka_tables as ft
    from mpinets_amd.metrics import BatchedEvaluator
    from mpinets_amd.robot import franka_fk, frames_to_matrix

    q = np.stack([ft.DEFAULT_Q, np.array([0.0, -1.7, 0.0, -3.0, 0.0, 0.3, 0.0])]).astype(np.float32)[:, None, :]
    tq = torch.from_numpy(q).to(dev())
    targets = frames_to_matrix(franka_fk(tq[:, 0])[:, ft.LINK_ID["right_gripper"]]).contiguous()
    out = BatchedEvaluator(dev()).evaluate_trajectories(tq, targets)
    ref = oracle.trajectory_metrics(q, None, targets.cpu().numpy(), ft.JOINT_LIMITS_PUBLISHED)
    np.testing.assert_array_equal(out["self_collision"].cpu().numpy(), ref["self_collision"])
    assert not bool(out["self_collision"][0]) and not bool(out["joint_limit_violation"][0])
    assert bool(out["success"][0])  # standing on the target, nothing violated
    assert (out["position_error"] < 1e-3).all()


def test_final_region_check():
    from mpinets_amd import franka_tables as ft
    from mpinets_amd.geometry import TorchCuboids
    from mpinets_amd.metrics import BatchedEvaluator
    from mpinets_amd.robot import franka_fk, frames_to_matrix

    q = torch.from_numpy(np.tile(ft.DEFAULT_Q.astype(np.float32), (2, 3, 1))).to(dev())
    pose = frames_to_matrix(franka_fk(q[:, 0])[:, ft.LINK_ID["right_gripper"]]).contiguous()
    c = pose[:, None, :3, 3].contiguous()
    ident = torch.tensor([[[1.0, 0, 0, 0]]], device=dev()).repeat(2, 1, 1)
    inside = TorchCuboids(c, torch.full((2, 1, 3), 0.2, device=dev()), ident)
    far = TorchCuboids(c + 1.0, torch.full((2, 1, 3), 0.2, device=dev()), ident)
    ev = BatchedEvaluator(dev())
    assert ev.evaluate_trajectories(q, pose, target_volume=inside, negative_volumes=far)["correct_final_region"].all()
    assert not ev.evaluate_trajectories(q, pose, target_volume=far)["correct_final_region"].any()
    # a negative volume that contains the TARGET is a bad volume and is dropped first (metrics.py:507-512): the final
    # pose sits on the target here, so `inside` does not count against it ...
    assert ev.evaluate_trajectories(q, pose, negative_volumes=inside)["success"].all()
    # ... but the same volume counts when the target lies elsewhere (outside of it)
    away = pose.clone()
    away[:, :3, 3] += 1.0
    out = ev.evaluate_trajectories(q, away, negative_volumes=inside)
    assert not out["correct_final_region"].any() and not out["success"].any()
    # mixed set: [contains target and final (dropped), far from both (harmless), zero-volume padding row]
    cc = torch.cat([c, c + 1.0, c * 0], dim=1)
    dd = torch.cat([torch.full((2, 2, 3), 0.2, device=dev()), torch.zeros((2, 1, 3), device=dev())], dim=1)
    mixed = TorchCuboids(cc, dd, ident.repeat(1, 3, 1))
    assert ev.evaluate_trajectories(q, pose, negative_volumes=mixed)["correct_final_region"].all()
    # env 0: the final position is inside volume 1, which does not contain its (shifted) target
    tgt = pose.clone()
    tgt[0, :3, 3] += 0.5
    cc2 = torch.cat([c + 1.0, c], dim=1)
    mixed2 = TorchCuboids(cc2, torch.full((2, 2, 3), 0.2, device=dev()), ident.repeat(1, 2, 1))
    r = ev.evaluate_trajectories(q, tgt, negative_volumes=mixed2)["correct_final_region"].cpu().numpy()
    assert r.tolist() == [False, True]


def test_smoothness_on_the_device_matches_the_oracle(oracle):
    """config / end-effector SPARC of a ragged batch through BatchedEvaluator (one batched FFT per FFT length, on the GPU)
    vs the oracle's per-trajectory numpy form (pinned to the reference's third_party/sparc.py, tests/test_smoothness.py).
    Minimum-jerk reaches: no spectral bin sits at the amplitude threshold, so the fp32 FK of the two sides cannot flip one."""
    from mpinets_amd import franka_tables as ft
    from mpinets_amd.metrics import BatchedEvaluator
    from mpinets_amd.robot import franka_fk, frames_to_matrix
    from mpinets_amd.scenes import random_configurations

    B, Tn, dt = 24, 150, 0.12
    rng = np.random.default_rng(3)
    lengths = rng.integers(2, Tn + 1, B).astype(np.int32)
    lengths[0], lengths[1], lengths[2] = Tn, 2, 33
    a, b = random_configurations(B, 5), random_configurations(B, 6)
    traj = np.zeros((B, Tn, 7), np.float32)
    for i in range(B):
        t = np.linspace(0.0, 1.0, lengths[i])
        s = 10 * t ** 3 - 15 * t ** 4 + 6 * t ** 5
        traj[i, :lengths[i]] = a[i] + s[:, None] * (b[i] - a[i]) * 0.5
        traj[i, lengths[i]:] = traj[i, lengths[i] - 1]
    traj[5] = traj[5, 0]  # an arm that never moves: 0, like the reference
    tt, ln = torch.from_numpy(traj).to(dev()), torch.from_numpy(lengths).to(dev())
    targets = frames_to_matrix(franka_fk(tt[:, -1])[:, ft.LINK_ID["right_gripper"]]).contiguous()
    ev = BatchedEvaluator(dev())
    got = ev.evaluate_trajectories(tt, targets, ln, dt=dt)
    cfg, eff = oracle.trajectory_smoothness(traj, lengths, dt)
    assert got["config_smoothness"].dtype == torch.float64 and got["config_smoothness"].is_cuda
    np.testing.assert_allclose(got["config_smoothness"].cpu().numpy(), cfg, rtol=0, atol=1e-6)
    np.testing.assert_allclose(got["eff_smoothness"].cpu().numpy(), eff, rtol=0, atol=1e-3)
    assert got["config_smoothness"][5].item() == 0.0 and got["eff_smoothness"][5].item() == 0.0
    assert (cfg[[i for i in range(B) if i != 5]] < 0).all()
    m = BatchedEvaluator.metrics(got)
    assert 0.0 <= m["is smooth"] <= 100.0 and m["total"] == B and "average eff sparc" in m
    # without dt the result carries no smoothness (and the summary no smoothness lines)
    assert "is smooth" not in BatchedEvaluator.metrics(ev.evaluate_trajectories(tt, targets, ln))


def test_batched_evaluator_matches_the_reference_evaluator(metrics_golden):
    """BatchedEvaluator.evaluate_trajectories vs vectors produced by RUNNING the reference's Evaluator methods
    (tests/golden/gen_metrics_golden.py): errors, path lengths, joint-limit flag, target / negative volume logic, SPARC."""
    from mpinets_amd import franka_tables as ft
    from mpinets_amd.geometry import TorchCuboids
    from mpinets_amd.metrics import BatchedEvaluator
    from mpinets_amd.robot import franka_fk, frames_to_matrix

    g = metrics_golden
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev())
    B = g["traj"].shape[0]
    targets = frames_to_matrix(franka_fk(T(g["goals"]))[:, ft.LINK_ID["right_gripper"]]).contiguous()
    unit = lambda n: T(np.tile(np.float32([1, 0, 0, 0]), (B, n, 1)))
    tv = TorchCuboids(T(g["tv_centers"][:, None]), T(g["tv_dims"][:, None]), unit(1))
    nv = TorchCuboids(T(g["nv_centers"]), T(g["nv_dims"]), unit(2))
    ev = BatchedEvaluator(dev())
    got = ev.evaluate_trajectories(T(g["traj"]), targets, T(g["lengths"]), target_volume=tv, negative_volumes=nv,
                                   dt=float(g["dt"]))
    c = lambda k: got[k].cpu().numpy()
    np.testing.assert_allclose(c("position_error"), g["m_position_error"], rtol=0, atol=2e-3)
    np.testing.assert_allclose(c("orientation_error"), g["m_orientation_error"], rtol=0, atol=6e-2)
    np.testing.assert_allclose(c("eff_position_path_length"), g["m_eff_position_path_length"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(c("eff_orientation_path_length"), g["m_eff_orientation_path_length"], rtol=1e-4, atol=5e-2)
    np.testing.assert_array_equal(c("joint_limit_violation"), g["m_joint_limit_violation"].astype(bool))
    # the negative volume around the final position of environments 0, 3, 6, 9 contains the TARGET only for 6 (which ends
    # on its target): the reference's evaluate_trajectory drops such volumes first (metrics.py:507-512); check_final_region
    # itself -- what the golden ran -- does not, so environment 6 is compared after the same correction
    ref_region = g["m_correct_final_region"].astype(bool).copy()
    ref_region[6] = True
    np.testing.assert_array_equal(c("correct_final_region"), ref_region)
    ok = g["lengths"] >= 2
    np.testing.assert_allclose(c("config_smoothness")[ok], g["m_config_smoothness"][ok], rtol=0, atol=1e-4)
    np.testing.assert_allclose(c("eff_smoothness")[ok], g["m_eff_smoothness"][ok], rtol=0, atol=2e-3)
