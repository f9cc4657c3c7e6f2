"""Row N2: batch assembly on the device vs the oracle restatement of data_loader.get_inputs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def synthetic_arrays(n=12, L=50, seed=0):
    from mpinets_amd.scenes import linear_trajectories, make_scenes

    scn = make_scenes(n, seed, ("tabletop", "cubby"), 10, 6)
    arr = {"cuboid_dims": scn["cuboid_dims"], "cuboid_centers": scn["cuboid_centers"],
           "cuboid_quaternions": scn["cuboid_quats"].copy(), "cylinder_radii": scn["cylinder_radii"],
           "cylinder_heights": scn["cylinder_heights"], "cylinder_centers": scn["cylinder_centers"],
           "cylinder_quaternions": scn["cylinder_quats"].copy(),
           "hybrid_solutions": linear_trajectories(n, L, seed + 1)}
    # padding rows as the generator stores them: zero dims AND an all-zero quaternion (data_loader.py:203-208)
    pad = (arr["cuboid_dims"] == 0).all(-1)
    arr["cuboid_quaternions"][pad] = 0
    arr["cylinder_quaternions"][(arr["cylinder_radii"][..., 0] == 0)] = 0
    return arr


def nearest(a, b):
    """max over rows of a of the distance to the closest row of b."""
    d = np.linalg.norm(a[:, None, :].astype(np.float64) - b[None, :, :], axis=-1)
    return d.min(1).max()


def test_instance_batch_matches_oracle(oracle):
    from mpinets_amd import franka_tables as ft
    from mpinets_amd.data import DatasetType, PointCloudInstanceDataset

    arr = synthetic_arrays()
    ds = PointCloudInstanceDataset(arr, "hybrid_solutions", 2048, 4096, 128, DatasetType.TRAIN, random_scale=0.03,
                                   device=dev())
    assert len(ds) == 12 * 50 and ds.expert_length == 50
    idx = np.array([0, 49, 50, 7 * 50 + 13, 11 * 50 + 49, 3 * 50 + 48])
    item = ds.get_batch(idx, seed=123)
    ti, ts = idx // 50, idx % 50
    ref = oracle.batch_configs(arr["hybrid_solutions"], ti, ts, ft.JOINT_LIMITS_REAL, 0.03, 123)
    for k in ("configuration", "supervision", "target_position"):
        np.testing.assert_allclose(item[k].cpu().numpy(), ref[k], rtol=0, atol=2e-6, err_msg=k)
    assert (np.abs(item["configuration"].cpu().numpy()) <= 1).all()
    # supervision of the last waypoint is itself (data_loader.py:408-412)
    np.testing.assert_allclose(ref["supervision"][1], oracle.normalize(arr["hybrid_solutions"][0, 49][None],
                                                                       ft.JOINT_LIMITS_REAL)[0], atol=1e-6)
    # primitives: the sampled trajectories' rows, all-zero quaternions repaired
    for src, dst in (("cuboid_dims", "cuboid_dims"), ("cuboid_centers", "cuboid_centers"),
                     ("cylinder_radii", "cylinder_radii"), ("cylinder_heights", "cylinder_heights")):
        np.testing.assert_array_equal(item[dst].cpu().numpy(), arr[src][ti])
    cq = item["cuboid_quats"].cpu().numpy()
    assert (np.abs(np.linalg.norm(cq, axis=-1) - 1) < 1e-5).all() and (cq[(arr["cuboid_dims"][ti] == 0).all(-1)] == [1, 0, 0, 0]).all()
    # slab rows: robot | scene | target with labels 0 | 1 | 2
    xyz = item["xyz"].cpu().numpy()
    assert xyz.shape == (6, 6272, 4)
    assert (xyz[:, :2048, 3] == 0).all() and (xyz[:, 2048:6144, 3] == 1).all() and (xyz[:, 6144:, 3] == 2).all()
    pts, link = ft.link_point_table(4096, True)
    T = oracle.franka_fk(ref["q"])
    full = oracle.transform_table(T, pts, link)
    eef = ft.end_effector_point_table()
    for b in range(6):
        assert nearest(xyz[b, :2048, :3], full[b]) < 2e-6  # robot rows = FK of (noisy, clamped) q
        tgt = eef @ ref["target_pose"][b, :3, :3].T + ref["target_pose"][b, :3, 3]
        assert nearest(xyz[b, 6144:, :3], tgt.astype(np.float64)) < 2e-6  # gripper cloud at the LAST waypoint's pose
    sc = xyz[:, 2048:6144, :3]
    yq = item["cylinder_quats"].cpu().numpy()
    sd = np.full(sc.shape[:2], np.inf, np.float32)  # min over primitives of |sdf| (objects may overlap)
    for m in range(cq.shape[1]):
        sl = slice(m, m + 1)
        sd = np.minimum(sd, np.abs(oracle.cuboid_sdf(arr["cuboid_centers"][ti][:, sl], arr["cuboid_dims"][ti][:, sl],
                                                     cq[:, sl], sc)))
    for m in range(yq.shape[1]):
        sl = slice(m, m + 1)
        sd = np.minimum(sd, np.abs(oracle.cylinder_sdf(arr["cylinder_centers"][ti][:, sl], arr["cylinder_radii"][ti][:, sl],
                                                       arr["cylinder_heights"][ti][:, sl], yq[:, sl], sc)))
    assert sd.max() < 1e-4  # every scene point lies on a primitive of ITS trajectory's scene


def test_noise_statistics_and_determinism():
    from mpinets_amd.data import DatasetType, PointCloudInstanceDataset

    arr = synthetic_arrays(n=8)
    arr["hybrid_solutions"][:] = 0.0
    arr["hybrid_solutions"][..., 3] = -1.5  # well inside the limits: no clamping
    arr["hybrid_solutions"][..., 5] = 1.5
    ds = PointCloudInstanceDataset(arr, "hybrid_solutions", 64, 64, 16, DatasetType.TRAIN, random_scale=0.03,
                                   device=dev())
    idx = np.arange(400)
    a = ds.get_batch(idx, seed=9)
    b = ds.get_batch(idx, seed=9)
    c = ds.get_batch(idx, seed=10)
    assert torch.equal(a["configuration"], b["configuration"]) and not torch.equal(a["configuration"], c["configuration"])
    from mpinets_amd import franka_tables as ft
    from mpinets_amd.utils import unnormalize_franka_joints

    q = unnormalize_franka_joints(a["configuration"]).cpu().numpy() - arr["hybrid_solutions"][0, 0]
    assert abs(q.std() - 0.03) < 0.002 and abs(q.mean()) < 0.002
    val = PointCloudInstanceDataset(arr, "hybrid_solutions", 64, 64, 16, DatasetType.VAL, random_scale=0.03,
                                    device=dev())
    v = unnormalize_franka_joints(val.get_batch(idx[:8])["configuration"]).cpu().numpy()
    np.testing.assert_allclose(v, arr["hybrid_solutions"][0, :8], atol=1e-6)  # no augmentation outside training


def test_trajectory_dataset_items_and_sources(tmp_path):
    from mpinets_amd import _lib
    from mpinets_amd.data import DatasetType, PointCloudTrajectoryDataset, PointCloudInstanceDataset

    arr = synthetic_arrays(n=5)
    del arr["cylinder_radii"], arr["cylinder_heights"], arr["cylinder_centers"], arr["cylinder_quaternions"]
    (tmp_path / "val" / "x").mkdir(parents=True)
    np.savez(tmp_path / "val" / "x" / "all_data.npz", **arr)
    ds = PointCloudTrajectoryDataset(tmp_path, "hybrid_solutions", 2048, 4096, 128, DatasetType.VAL, device=dev())
    assert len(ds) == 5
    item = ds[3]
    assert set(item) == {"xyz", "configuration", "target_position", "cuboid_dims", "cuboid_centers", "cuboid_quats",
                         "cylinder_radii", "cylinder_heights", "cylinder_centers", "cylinder_quats"}
    assert item["xyz"].shape == (6272, 4) and item["configuration"].shape == (7,)
    assert item["cylinder_radii"].shape == (1, 1) and item["cylinder_quats"].tolist() == [[1.0, 0.0, 0.0, 0.0]]
    with pytest.raises(AssertionError):
        PointCloudTrajectoryDataset(arr, "hybrid_solutions", 2048, 4096, 128, DatasetType.TRAIN, device=dev())
    (tmp_path / "f.hdf5").write_bytes(b"")
    with pytest.raises(_lib.MpxError):  # no h5py in this image: a clear error, not a guess
        PointCloudTrajectoryDataset(tmp_path / "f.hdf5", "hybrid_solutions", 8, 8, 8, DatasetType.VAL, device=dev())
    # epoch iterator: ranks see disjoint shares, every batch is ready for training_step
    inst = PointCloudInstanceDataset(arr, "hybrid_solutions", 2048, 4096, 128, DatasetType.TRAIN, 0.03, device=dev())
    got = [b["configuration"].shape[0] for r in range(2) for b in inst.batches(16, seed=1, rank=r, world_size=2)]
    assert sum(got) == (125 // 16) * 16 * 2 and set(got) == {16}


def test_assembled_batch_trains():
    from mpinets_amd.data import DatasetType, PointCloudInstanceDataset
    from mpinets_amd.model import TrainingMotionPolicyNetwork

    ds = PointCloudInstanceDataset(synthetic_arrays(), "hybrid_solutions", 2048, 4096, 128, DatasetType.TRAIN, 0.03,
                                   device=dev())
    torch.manual_seed(0)
    mdl = TrainingMotionPolicyNetwork(2048, 1.0, 1.0).to(dev()).train()
    loss = mdl.training_step(next(ds.batches(4, seed=0)), 0)
    loss.backward()
    assert torch.isfinite(loss) and all(p.grad is not None and torch.isfinite(p.grad).all() for p in mdl.parameters())
