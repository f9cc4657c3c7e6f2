"""GPU: the drop-in callers' entry points vs vectors produced by RUNNING the reference's data_loader.py and
run_inference.py (tests/golden/gen_caller_golden.py).  Rows a11 (slab assembly), a15 (rollout_until_success), N2
(get_inputs / dataset classes)."""
import random

import numpy as np
import pytest
import torch

from test_gpu_caller_replay import Target, replay_make_point_cloud_from_primitives
from test_oracle_callers import NR, NS, NT, _obstacles
from test_oracle_model import golden_state_dict

pytestmark = pytest.mark.gpu
TOL = 1e-5


def dev():
    return torch.device("cuda:0")


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


def _arrays(g):
    return {k[2:]: g[k] for k in g if k.startswith("d_")}


def _on_surface(points, item):
    """max |sdf| of the points to the union of the item's unmasked primitives (the drop-in geometry classes)."""
    from mpinets_amd.geometry import TorchCuboids, TorchCylinders

    b = lambda k: item[k].reshape((1,) + tuple(item[k].shape)).float().to(dev())
    cub = TorchCuboids(b("cuboid_centers"), b("cuboid_dims"), b("cuboid_quats"))
    cyl = TorchCylinders(b("cylinder_centers"), b("cylinder_radii"), b("cylinder_heights"), b("cylinder_quats"))
    p = points.reshape(1, -1, 3).float().to(dev())
    return float(torch.minimum(cub.sdf(p), cyl.sdf(p)).abs().max())


def test_validation_dataset_item(caller_golden):
    """PointCloudTrajectoryDataset[2] on the device == the reference's item: every deterministic field within 1e-5 (the
    robot / target rows with the reference's column subsets), the scene rows on the same surfaces."""
    from mpinets_amd.data import DatasetType, PointCloudTrajectoryDataset

    g = caller_golden
    ds = PointCloudTrajectoryDataset(_arrays(g), "hybrid_solutions", NR, NS, NT, DatasetType.VAL, device=dev())
    assert len(ds) == 4 and ds.expert_length == 50
    item = ds.get_inputs_batch(torch.tensor([2]), None, with_supervision=False, seed=5,
                               robot_subset=T(g["dv_robot_subset"]), target_subset=T(g["dv_target_subset"]))
    item = {k: v[0] for k, v in item.items()}
    assert set(item) == {k[3:] for k in g if k.startswith("dv_") and not k.endswith("_subset")}
    for k in ("configuration", "target_position", "cuboid_dims", "cuboid_centers", "cuboid_quats", "cylinder_radii",
              "cylinder_heights", "cylinder_centers", "cylinder_quats"):
        assert tuple(item[k].shape) == g["dv_" + k].shape, k
        np.testing.assert_allclose(item[k].cpu().numpy(), g["dv_" + k], rtol=0, atol=TOL, err_msg=k)
    xyz, ref = item["xyz"].cpu().numpy(), g["dv_xyz"]
    np.testing.assert_array_equal(xyz[:, 3], ref[:, 3])
    np.testing.assert_allclose(xyz[:NR, :3], ref[:NR, :3], rtol=0, atol=TOL)
    np.testing.assert_allclose(xyz[NR + NS:, :3], ref[NR + NS:, :3], rtol=0, atol=TOL)
    # the scene rows come from different random streams (device Philox vs np.random): same surfaces, same extent
    assert _on_surface(item["xyz"][NR:NR + NS, :3], item) < 2e-5 and _on_surface(T(ref[NR:NR + NS, :3]), item) < 2e-5
    np.testing.assert_allclose(xyz[NR:NR + NS, :3].min(0), ref[NR:NR + NS, :3].min(0), atol=0.03)
    np.testing.assert_allclose(xyz[NR:NR + NS, :3].max(0), ref[NR:NR + NS, :3].max(0), atol=0.03)


def test_dataset_without_cylinders_and_without_the_primitive_axis(caller_golden):
    from mpinets_amd.data import DatasetType, PointCloudTrajectoryDataset

    g = caller_golden
    arr = {"cuboid_dims": g["d_cuboid_dims"][:, 0], "cuboid_centers": g["d_cuboid_centers"][:, 0],
           "cuboid_quaternions": g["d_cuboid_quaternions"][:, 0], "hybrid_solutions": g["d_hybrid_solutions"]}
    ds = PointCloudTrajectoryDataset(arr, "hybrid_solutions", NR, NS, NT, DatasetType.VAL, device=dev())
    item = ds[1]
    for k in ("cuboid_dims", "cuboid_centers", "cuboid_quats", "cylinder_radii", "cylinder_heights", "cylinder_centers",
              "cylinder_quats"):
        assert tuple(item[k].shape) == g["d1_" + k].shape, k
        np.testing.assert_allclose(item[k].cpu().numpy(), g["d1_" + k], rtol=0, atol=TOL, err_msg=k)
    np.testing.assert_allclose(item["configuration"].cpu().numpy(), g["d1_configuration"], rtol=0, atol=TOL)
    np.testing.assert_allclose(item["target_position"].cpu().numpy(), g["d1_target_position"], rtol=0, atol=TOL)


def test_training_dataset_item_without_noise(caller_golden, oracle):
    """PointCloudInstanceDataset: index -> (trajectory, waypoint), the last waypoint supervised by itself; with the noise
    scale at zero the configuration is the clamped, normalised waypoint (the noisy case: tests/test_oracle_callers.py
    for the reference's arithmetic, tests/test_gpu_data.py for the device draw)."""
    from mpinets_amd import franka_tables as ft
    from mpinets_amd.data import DatasetType, PointCloudInstanceDataset

    g = caller_golden
    ds = PointCloudInstanceDataset(_arrays(g), "hybrid_solutions", NR, NS, NT, DatasetType.TRAIN, random_scale=0.0, device=dev())
    assert len(ds) == 200
    item = ds[3 * 50 + 49]
    np.testing.assert_allclose(item["supervision"].cpu().numpy(), g["dt_supervision"], rtol=0, atol=TOL)
    np.testing.assert_allclose(item["configuration"].cpu().numpy(),
                               oracle.normalize(g["d_hybrid_solutions"][3, 49][None], ft.JOINT_LIMITS_REAL)[0], rtol=0, atol=TOL)
    np.testing.assert_allclose(item["target_position"].cpu().numpy(), g["dt_target_position"], rtol=0, atol=TOL)
    for k in ("cuboid_quats", "cylinder_quats", "cuboid_dims"):
        np.testing.assert_allclose(item[k].cpu().numpy(), g["dt_" + k], rtol=0, atol=TOL)


@pytest.fixture(scope="module")
def policy(model_golden):
    from mpinets_amd.model import MotionPolicyNetwork

    m = MotionPolicyNetwork().eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in golden_state_dict(model_golden).items()}, strict=True)
    return m.to(dev())


def test_make_point_cloud_from_primitives(caller_golden):
    """The driver's slab assembly (run_inference.py:93-134) over this repo's sampler, same seeds: scene rows bit for bit."""
    from mpinets_amd.robot import FrankaSampler

    g = caller_golden
    obstacles = _obstacles(g["i_cuboid_centers"], g["i_cuboid_dims"], g["i_cuboid_quats"], g["i_cylinder_centers"],
                           g["i_cylinder_radii"], g["i_cylinder_heights"], g["i_cylinder_quats"])
    random.seed(61), np.random.seed(61)
    with torch.no_grad():
        pc = replay_make_point_cloud_from_primitives(torch.as_tensor(g["i_q0"]).unsqueeze(0), Target(g["i_target0"]), obstacles,
                                                     FrankaSampler("cpu", use_cache=True))
    pc, ref = pc.numpy(), g["i_slab"]
    np.testing.assert_array_equal(pc[:, 3], ref[:, 3])
    np.testing.assert_array_equal(pc[NR:NR + NS, :3], ref[NR:NR + NS, :3])
    np.testing.assert_allclose(pc[:, :3], ref[:, :3], rtol=0, atol=TOL)


@pytest.mark.parametrize("which,target_key,length", [("full", "i_target0", 13), ("stop", "i_target1", 8)])
def test_rollout_until_success(caller_golden, policy, monkeypatch, which, target_key, length):
    """mpinets_amd.rollout.rollout_until_success (the reference's signature) == the reference's run: trajectory, the
    column subsets it drew, the slab it left behind and where it left np.random."""
    from mpinets_amd.robot import FrankaSampler
    from mpinets_amd.rollout import rollout_until_success

    g = caller_golden
    slab = T(g["i_slab"][None].copy())
    drawn = []
    real = FrankaSampler._draw

    def spy(self, n, total=None):
        s = real(self, n, total)
        drawn.append(s.cpu().numpy())
        return s

    monkeypatch.setattr(FrankaSampler, "_draw", spy)
    np.random.seed(62)
    with torch.no_grad():
        traj = rollout_until_success(policy, g["i_q0"], Target(g[target_key]), slab, FrankaSampler(dev(), use_cache=True),
                                     max_rollout_length=12)
    assert traj.shape == (length, 7)
    err = np.abs(traj - g[f"i_traj_{which}"]).max()
    print(f"rollout_until_success ({which}) vs reference-run golden: {err:.2e}")
    assert err <= 5 * TOL
    subs = g[f"i_subsets_{which}"]
    np.testing.assert_array_equal(np.stack(drawn)[:len(subs)], subs)  # (a final draw is undone when the loop stops early)
    st = np.random.get_state()
    np.testing.assert_array_equal(np.concatenate((st[1][:8].astype(np.int64), [st[2]])), g[f"i_rng_{which}"])
    np.testing.assert_allclose(slab[0, :NR, :3].cpu().numpy(), g[f"i_robot_{which}"], rtol=0, atol=5 * TOL)
    np.testing.assert_array_equal(slab[0, NR:].cpu().numpy(), g["i_slab"][NR:])
