"""FK, robot cloud, collision spheres and the fused collision check vs the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


def test_sincos_accuracy_on_cpu_side(oracle):
    x = np.linspace(-6.5, 6.5, 20001).astype(np.float32)
    s, c = oracle.sincos(x)
    assert np.abs(s - np.sin(x.astype(np.float64))).max() < 3e-7
    assert np.abs(c - np.cos(x.astype(np.float64))).max() < 3e-7


def test_fk_frames_match_oracle(oracle):
    from mpinets_amd import franka_tables as ft
    from mpinets_amd.robot import franka_fk
    from mpinets_amd.scenes import random_configurations

    q = random_configurations(777, 1)
    q[0] = ft.DEFAULT_Q
    got = franka_fk(T(q)).cpu().numpy()
    ref = oracle.franka_fk(q)
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-6)
    print("FK bitwise-equal fraction: %.5f" % (got == ref).mean())
    # rotations are orthonormal, flange sits 0.107 above link7 along its z axis
    R = got[:, :, :9].reshape(-1, 15, 3, 3)
    assert np.abs(R @ np.swapaxes(R, -1, -2) - np.eye(3)).max() < 1e-5
    d = got[:, 8, 9:] - got[:, 7, 9:]
    np.testing.assert_allclose(np.linalg.norm(d, axis=1), 0.107, atol=1e-6)


def test_robot_cloud_and_inplace_slab(oracle):
    from mpinets_amd.robot import FrankaSampler
    from mpinets_amd.scenes import random_configurations

    np.random.seed(3)
    smp = FrankaSampler(dev())
    q = random_configurations(37, 2)
    subset = smp.draw_subset(2048)
    slab = torch.full((37, 6272, 4), 7.0, device=dev())
    smp.sample_into(T(q), slab, subset)
    ref = oracle.transform_table(oracle.franka_fk(q), smp.table_pts.cpu().numpy(), smp.table_link.cpu().numpy(),
                                 subset.cpu().numpy())
    np.testing.assert_allclose(slab[:, :2048, :3].cpu().numpy(), ref, rtol=0, atol=1e-6)
    assert (slab[:, :2048, 3] == 7).all() and (slab[:, 2048:] == 7).all()  # nothing else touched
    full = smp.sample(T(q))
    assert full.shape == (37, 4096, 3)
    np.random.seed(5)
    a = smp.sample(T(q), 1024)
    np.random.seed(5)
    b = smp.sample(T(q), 1024)
    assert torch.equal(a, b) and a.shape == (37, 1024, 3)


def test_end_effector_cloud_and_pose(oracle):
    from mpinets_amd import franka_tables as ft
    from mpinets_amd.robot import FrankaSampler
    from mpinets_amd.scenes import random_configurations

    np.random.seed(0)
    smp = FrankaSampler(dev())
    q = random_configurations(9, 4)
    pose = smp.end_effector_pose(T(q))
    ref = oracle.frames_to_4x4(oracle.franka_fk(q)[:, ft.LINK_ID["right_gripper"]])
    np.testing.assert_allclose(pose.cpu().numpy(), ref, atol=1e-6)
    pts = smp.sample_end_effector(pose, 128)
    assert pts.shape == (9, 128, 3)
    # gripper points expressed in the right_gripper frame stay within the hand's extent
    local = torch.einsum("bij,bpj->bpi", pose[:, :3, :3].transpose(1, 2), pts - pose[:, None, :3, 3])
    assert local.abs().max() < 0.2


def test_collision_spheres_grouping(oracle):
    from mpinets_amd.robot import FrankaCollisionSampler
    from mpinets_amd.scenes import random_configurations

    cs = FrankaCollisionSampler(dev(), with_base_link=False)
    assert cs.num_spheres == 56
    q = random_configurations(11, 6)
    groups = cs.compute_spheres(T(q))
    assert len(groups) == 9 and sum(g[1].shape[1] for g in groups) == 56
    ref = oracle.transform_table(oracle.franka_fk(q), cs.centers.cpu().numpy(), cs.links.cpu().numpy())
    got = torch.cat([g[1] for g in groups], dim=1).cpu().numpy()
    np.testing.assert_allclose(got, ref, atol=1e-6)
    assert FrankaCollisionSampler(dev(), with_base_link=True).num_spheres == 57


@pytest.mark.parametrize("B,Tn,M1,M2", [(5, 1, 16, 16), (3, 50, 16, 16), (33, 7, 16, 16), (4, 130, 40, 16), (2, 65, 64, 64),
                                        (3, 9, 70, 16), (2, 1, 16, 65)])
def test_fused_collision_check_matches_oracle(oracle, B, Tn, M1, M2):
    """(M1, M2 <= 64: the per-environment kernel -- masks as two scalar words, waypoint chunks of 64, pairs flattened over
    the lanes; above: the general kernel.  Both are bit-identical in their minima.)"""
    from mpinets_amd.geometry import TorchCuboids, TorchCylinders
    from mpinets_amd.robot import FrankaCollisionSampler
    from mpinets_amd.scenes import linear_trajectories, make_scenes

    scn = make_scenes(B, 11, ("tabletop", "cubby", "dresser") if M1 >= 40 else ("tabletop", "cubby"), M1, M2)
    traj = linear_trajectories(B, Tn, 9)
    cs = FrankaCollisionSampler(dev())
    cub = TorchCuboids(T(scn["cuboid_centers"]), T(scn["cuboid_dims"]), T(scn["cuboid_quats"]))
    cyl = TorchCylinders(T(scn["cylinder_centers"]), T(scn["cylinder_radii"]), T(scn["cylinder_heights"]),
                         T(scn["cylinder_quats"]))
    flags, msdf = cs.check(T(traj), cub, cyl, return_sdf=True)
    frames = oracle.franka_fk(traj.reshape(-1, 7))
    centres = oracle.transform_table(frames, cs.centers.cpu().numpy(), cs.links.cpu().numpy()).reshape(B, Tn, 56, 3)
    oflags, omsdf = oracle.collision_flags(
        centres, cs.radii.cpu().numpy(), (scn["cuboid_centers"], scn["cuboid_dims"], scn["cuboid_quats"]),
        (scn["cylinder_centers"], scn["cylinder_radii"], scn["cylinder_heights"], scn["cylinder_quats"]))
    np.testing.assert_allclose(msdf.cpu().numpy(), omsdf, rtol=0, atol=1e-5)
    border = np.abs(omsdf - cs.radii.cpu().numpy()[None, None]).min(axis=(1, 2)) < 1e-5
    np.testing.assert_array_equal(flags.cpu().numpy()[~border], oflags[~border])
    # same answer through the reference's unfused formulation (model.py:301-312)
    has = torch.zeros(B, dtype=torch.bool, device=dev())
    for radius, spheres in cs.compute_spheres(T(traj.reshape(-1, 7))):
        seq = spheres.reshape(B, -1, spheres.shape[-2], 3)
        sdf = torch.minimum(cub.sdf_sequence(seq), cyl.sdf_sequence(seq))
        has |= torch.any(sdf.reshape(B, -1) <= radius, dim=-1)
    np.testing.assert_array_equal(has.cpu().numpy()[~border], oflags[~border])
    # no primitives at all -> nothing collides
    assert not cs.check(T(traj), None, None).any()


def test_collision_check_accepts_unaligned_frame_pointers():
    """the per-environment kernel stages the frames as float4 rows; a caller whose inv_frames view starts at
    an odd storage offset (raw C callers, torch slices) must get the same flags and distances, not a misaligned-access
    fault: such pointers take the general kernel (include/mpinets_hip.h, mpx_franka_collision)."""
    from mpinets_amd.geometry import TorchCuboids, TorchCylinders
    from mpinets_amd.robot import FrankaCollisionSampler
    from mpinets_amd.scenes import linear_trajectories, make_scenes

    B, Tn = 6, 9
    scn = make_scenes(B, 5, ("tabletop", "cubby"), 16, 16)
    traj = T(linear_trajectories(B, Tn, 2))
    cs = FrankaCollisionSampler(dev())
    cub = TorchCuboids(T(scn["cuboid_centers"]), T(scn["cuboid_dims"]), T(scn["cuboid_quats"]))
    cyl = TorchCylinders(T(scn["cylinder_centers"]), T(scn["cylinder_radii"]), T(scn["cylinder_heights"]), T(scn["cylinder_quats"]))
    flags, msdf = cs.check(traj, cub, cyl, return_sdf=True)
    for prim in (cub, cyl):  # the same frames, one float further into a larger buffer
        buf = torch.empty(prim.inv_frames.numel() + 1, dtype=torch.float32, device=dev())
        shifted = buf[1:].view_as(prim.inv_frames)
        shifted.copy_(prim.inv_frames)
        assert shifted.data_ptr() % 16 == 4
        prim.inv_frames = shifted
    flags2, msdf2 = cs.check(traj, cub, cyl, return_sdf=True)
    assert torch.equal(flags, flags2) and torch.equal(msdf, msdf2)


def test_flags_only_sweep_decides_like_the_distance_form_at_the_boundary():
    """The flags-only form of the per-environment kernel skips the square root behind two conservative bounds and falls
    back to the exact arithmetic in between: spheres whose radius IS their distance (and the floats just below / above
    it, and radii 1e-6 off) must get the flag `min_sdf <= radius` gives -- bit for bit, never a bound's guess."""
    from mpinets_amd import _lib
    from mpinets_amd.geometry import TorchCuboids, TorchCylinders
    from mpinets_amd.robot import FrankaCollisionSampler
    from mpinets_amd.scenes import linear_trajectories, make_scenes

    B, Tn = 24, 3
    scn = make_scenes(B, 17, ("tabletop", "cubby", "dresser"), 40, 16)
    traj = T(linear_trajectories(B, Tn, 4))
    cs = FrankaCollisionSampler(dev())
    cub = TorchCuboids(T(scn["cuboid_centers"]), T(scn["cuboid_dims"]), T(scn["cuboid_quats"]))
    cyl = TorchCylinders(T(scn["cylinder_centers"]), T(scn["cylinder_radii"]), T(scn["cylinder_heights"]),
                         T(scn["cylinder_quats"]))
    _, msdf = cs.check(traj, cub, cyl, return_sdf=True)  # [B,T,56]
    msdf = msdf.cpu().numpy()
    yr, yh = cyl.radii.contiguous(), cyl.heights.contiguous()
    cd = cub.dims.contiguous()

    def flags_only(radii, b, t):  # one (environment, waypoint) with its own sphere radii
        fl = torch.zeros(1, dtype=torch.int32, device=dev())
        _lib.call("mpx_franka_collision", _lib.ptr(traj[b, t:t + 1].contiguous()), 1, 1, cs.finger, _lib.ptr(cs.centers),
                  _lib.ptr(T(radii)), _lib.ptr(cs.links), 56, _lib.ptr(cub.inv_frames[b:b + 1].contiguous()),
                  _lib.ptr(cd[b:b + 1].contiguous()), 40, _lib.ptr(cyl.inv_frames[b:b + 1].contiguous()),
                  _lib.ptr(yr[b:b + 1].contiguous()), _lib.ptr(yh[b:b + 1].contiguous()), 16, _lib.ptr(fl), None)
        return bool(fl.item())

    checked = 0
    for b in range(B):
        for t in range(Tn):
            d = msdf[b, t]
            if not (d > 1e-3).all():
                continue  # (a sphere inside an obstacle: every non-negative radius collides, nothing to decide)
            s = int(np.argmin(d))
            for radii, want in ((np.full(56, -1.0), False),):
                assert flags_only(radii.astype(np.float32), b, t) is want
            base = np.zeros(56, np.float32)  # every other sphere: radius 0 < its distance
            for r, want in ((d[s], True), (np.nextafter(d[s], np.float32(0)), False), (np.nextafter(d[s], np.float32(9)), True),
                            (d[s] * np.float32(1 - 1e-6), False), (d[s] * np.float32(1 + 1e-6), True),
                            (d[s] * np.float32(0.99), False), (d[s] * np.float32(1.01), True)):
                radii = base.copy()
                radii[s] = r
                assert flags_only(radii, b, t) is want, (b, t, s, float(d[s]), float(r))
            checked += 1
    assert checked >= 10


def test_joint_step(oracle):
    from mpinets_amd import _lib
    from mpinets_amd import franka_tables as ft

    rng = np.random.default_rng(2)
    qn = rng.uniform(-1, 1, (100, 7)).astype(np.float32)
    dq = rng.normal(0, 0.5, (100, 7)).astype(np.float32)
    lim = T(ft.JOINT_LIMITS_REAL.astype(np.float32))
    a, b = torch.empty(100, 7, device=dev()), torch.empty(100, 7, device=dev())
    tq, td = T(qn), T(dq)
    _lib.call("mpx_joint_step", _lib.ptr(tq), _lib.ptr(td), _lib.ptr(lim), 100, _lib.ptr(a), _lib.ptr(b), None)
    ref_n = np.clip(qn + dq, -1, 1)
    np.testing.assert_array_equal(a.cpu().numpy(), ref_n)
    np.testing.assert_allclose(b.cpu().numpy(), oracle.unnormalize(ref_n, ft.JOINT_LIMITS_REAL), atol=1e-6)


@pytest.mark.parametrize("total,n,seed,draw", [(4096, 2048, 0, 0), (4096, 2048, 12345678901, 49), (100, 100, 3, 1),
                                               (5000, 1, 7, 2), (4099, 4096, 1, 5), (64, 16, 9, 0)])
def test_device_subset_draw_equals_oracle(oracle, total, n, seed, draw):
    """mpx_draw_subset (radix select of the n smallest Philox keys + ordering, one workgroup) == the oracle's
    sort-all-keys restatement, index for index."""
    from mpinets_amd import _lib

    out = torch.full((n,), -1, dtype=torch.int32, device="cuda:0")
    _lib.call("mpx_draw_subset", total, n, seed, draw, _lib.ptr(out))
    np.testing.assert_array_equal(out.cpu().numpy(), oracle.draw_subset(total, n, seed, draw))
    lib = _lib.load()
    assert lib.mpx_draw_subset(10, 11, 0, 0, out.data_ptr(), None) != 0  # more rows than the table has
    assert lib.mpx_draw_subset(9000, 5000, 0, 0, out.data_ptr(), None) != 0 and b"4096" in lib.mpx_last_error()
