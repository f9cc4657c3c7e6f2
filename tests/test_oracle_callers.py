"""CPU: host logic + oracle vs vectors produced by RUNNING the reference's data_loader.py and run_inference.py
(tests/golden/gen_caller_golden.py): ``get_inputs`` (zero-quaternion repair, dummy cylinder, single-primitive rows,
normalisation, noise clamp, slab layout, scene cloud under the same seeds) and the inference driver
(``make_point_cloud_from_primitives``, ``rollout_until_success`` incl. its early stop before the resample)."""
import random

import numpy as np
import pytest

from test_oracle_model import golden_state_dict

NR, NS, NT = 2048, 4096, 128
TOL = 1e-6


@pytest.fixture(scope="module")
def g(caller_golden):
    return caller_golden


def _tables():
    from mpinets_amd import franka_tables as ft

    return ft.link_point_table(4096, True), ft.end_effector_point_table(), ft.JOINT_LIMITS_REAL


def _obstacles(cc, cd, cq, yc, yr, yh, yq):
    """The zero-volume filter + object construction of data_loader.py:237-256 with this repo's primitives."""
    from mpinets_amd.primitives import Cuboid, Cylinder

    cub = [Cuboid(c, d, q) for c, d, q in zip(cc, cd, cq)]
    cyl = [Cylinder(c, r, h, q) for c, r, h, q in zip(yc, yr.squeeze(1), yh.squeeze(1), yq)]
    return [c for c in cub if not c.is_zero_volume()] + [c for c in cyl if not c.is_zero_volume()]


def _pose_points(pose, eef, subset):
    return (np.einsum("ij,nj->ni", pose[:3, :3], eef[subset]) + pose[None, :3, 3]).astype(np.float32)


def test_get_inputs_validation_item_reproduced_under_the_same_seeds(oracle, g):
    """PointCloudTrajectoryDataset[2] (data_loader.py:141-280,325-342): every field, the scene cloud bit for bit."""
    from mpinets_amd.geometry import construct_mixed_point_cloud

    (pts, link), eef, lim = _tables()
    traj = g["d_hybrid_solutions"]
    cq, yq = oracle.repair_quaternions(g["d_cuboid_quaternions"][2]), oracle.repair_quaternions(g["d_cylinder_quaternions"][2])
    np.testing.assert_array_equal(g["dv_cuboid_quats"], cq)
    np.testing.assert_array_equal(g["dv_cylinder_quats"], yq)
    pad = (g["d_cuboid_dims"][2] == 0).all(-1)
    assert pad.any() and (g["dv_cuboid_quats"][pad] == [1, 0, 0, 0]).all()  # all-zero quaternions became unit
    for k in ("cuboid_dims", "cuboid_centers", "cylinder_radii", "cylinder_heights", "cylinder_centers"):
        np.testing.assert_array_equal(g["dv_" + k], g["d_" + k][2])
    np.testing.assert_allclose(g["dv_configuration"], oracle.normalize(traj[2, 0][None], lim)[0], rtol=0, atol=TOL)
    pose = oracle.frames_to_4x4(oracle.franka_fk(traj[2, -1][None])[0, oracle.RIGHT_GRIPPER_FRAME])
    np.testing.assert_allclose(g["dv_target_position"], pose[:3, 3], rtol=0, atol=TOL)
    # the reference's draw order: target subset, robot subset, then construct_mixed_point_cloud
    random.seed(41), np.random.seed(41)
    tsub = np.random.choice(len(eef), NT, replace=False)
    rsub = np.random.choice(len(pts), NR, replace=False)
    np.testing.assert_array_equal(tsub, g["dv_target_subset"])
    np.testing.assert_array_equal(rsub, g["dv_robot_subset"])
    obstacles = _obstacles(g["dv_cuboid_centers"], g["dv_cuboid_dims"], cq, g["dv_cylinder_centers"], g["dv_cylinder_radii"],
                           g["dv_cylinder_heights"], yq)
    assert 0 < len(obstacles) < len(cq) + len(yq)
    cloud = construct_mixed_point_cloud(obstacles, NS)
    xyz = g["dv_xyz"]
    np.testing.assert_array_equal(xyz[NR:NR + NS, :3], cloud[:, :3].astype(np.float32))
    np.testing.assert_array_equal(xyz[:, 3], np.repeat([0, 1, 2], [NR, NS, NT]).astype(np.float32))
    robot = oracle.transform_table(oracle.franka_fk(traj[2, 0][None]), pts, link, rsub.astype(np.int32))[0]
    np.testing.assert_allclose(xyz[:NR, :3], robot, rtol=0, atol=TOL)
    np.testing.assert_allclose(xyz[NR + NS:, :3], _pose_points(pose, eef, tsub), rtol=0, atol=TOL)


def test_get_inputs_single_cuboid_rows_and_dummy_cylinder(g):
    """A file whose primitive arrays have no M axis and no cylinders at all (data_loader.py:189-201, 210-215)."""
    assert g["d1_cuboid_dims"].shape == (1, 3) and g["d1_cuboid_centers"].shape == (1, 3) and g["d1_cuboid_quats"].shape == (1, 4)
    np.testing.assert_array_equal(g["d1_cuboid_dims"][0], g["d_cuboid_dims"][1, 0])
    assert g["d1_cylinder_radii"].shape == (1, 1) and g["d1_cylinder_heights"].shape == (1, 1)
    assert not g["d1_cylinder_radii"].any() and not g["d1_cylinder_heights"].any() and not g["d1_cylinder_centers"].any()
    np.testing.assert_array_equal(g["d1_cylinder_quats"], [[1.0, 0, 0, 0]])
    assert g["d1_xyz"].shape == (NR + NS + NT, 4)


def test_get_inputs_training_item_noise_clamp_and_supervision(oracle, g):
    """PointCloudInstanceDataset[3*50+49] (data_loader.py:166-183, 398-417): noise, clamp to the limits, normalise; the
    last waypoint is supervised by itself."""
    (pts, link), _, lim = _tables()
    traj = g["d_hybrid_solutions"]
    lim32 = lim.astype(np.float32)
    noisy = (np.float32(0.015) * g["dt_noise"] + traj[3, 49]).astype(np.float32)
    noisy = np.minimum(np.maximum(noisy, lim32[:, 0]), lim32[:, 1])
    np.testing.assert_allclose(g["dt_configuration"], oracle.normalize(noisy[None], lim)[0], rtol=0, atol=TOL)
    assert (np.abs(g["dt_configuration"]) <= 1).all()
    np.testing.assert_allclose(g["dt_supervision"], oracle.normalize(traj[3, 49][None], lim)[0], rtol=0, atol=TOL)
    robot = oracle.transform_table(oracle.franka_fk(noisy[None]), pts, link, g["dt_robot_subset"])[0]
    np.testing.assert_allclose(g["dt_xyz"][:NR, :3], robot, rtol=0, atol=TOL)  # the cloud is sampled at the NOISY joints


def test_make_point_cloud_from_primitives_reproduced_under_the_same_seeds(oracle, g):
    """run_inference.py:93-134: scene cloud first, then the robot cloud, then the target cloud."""
    from mpinets_amd.geometry import construct_mixed_point_cloud

    (pts, link), eef, _ = _tables()
    obstacles = _obstacles(g["i_cuboid_centers"], g["i_cuboid_dims"], g["i_cuboid_quats"], g["i_cylinder_centers"],
                           g["i_cylinder_radii"], g["i_cylinder_heights"], g["i_cylinder_quats"])
    random.seed(61), np.random.seed(61)
    cloud = construct_mixed_point_cloud(obstacles, NS)
    rsub = np.random.choice(len(pts), NR, replace=False)
    tsub = np.random.choice(len(eef), NT, replace=False)
    np.testing.assert_array_equal(rsub, g["i_slab_robot_subset"])
    np.testing.assert_array_equal(tsub, g["i_slab_target_subset"])
    slab = g["i_slab"]
    np.testing.assert_array_equal(slab[NR:NR + NS, :3], cloud[:, :3].astype(np.float32))
    np.testing.assert_array_equal(slab[:, 3], np.repeat([0, 1, 2], [NR, NS, NT]).astype(np.float32))
    robot = oracle.transform_table(oracle.franka_fk(g["i_q0"][None]), pts, link, rsub.astype(np.int32))[0]
    np.testing.assert_allclose(slab[:NR, :3], robot, rtol=0, atol=TOL)
    np.testing.assert_allclose(slab[NR + NS:, :3], _pose_points(g["i_target0"], eef, tsub), rtol=0, atol=TOL)


@pytest.mark.parametrize("which,target_key,length", [("full", "i_target0", 13), ("stop", "i_target1", 8)])
def test_rollout_until_success_loop(oracle, g, model_golden, which, target_key, length):
    """run_inference.py:137-191: the full-length run and the early stop (success test before the resample)."""
    (pts, link), _, lim = _tables()
    sd = golden_state_dict(model_golden)
    assert str(g["param_sha256"]) == str(model_golden["param_sha256"])
    subsets = g[f"i_subsets_{which}"]
    used = []

    def sampler(qt, i):
        used.append(i)
        return oracle.transform_table(oracle.franka_fk(qt), pts, link, subsets[i])

    slab = g["i_slab"][None].copy()
    traj = oracle.rollout_until_success(sd, g["i_q0"], g[target_key], slab, sampler, lim, max_rollout_length=12)
    assert traj.shape == (length, 7) and len(used) == len(subsets) == (12 if which == "full" else length - 2)
    np.testing.assert_allclose(traj, g[f"i_traj_{which}"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(slab[0, :NR, :3], g[f"i_robot_{which}"], rtol=0, atol=1e-5)
    np.testing.assert_array_equal(slab[0, NR:], g["i_slab"][NR:])
