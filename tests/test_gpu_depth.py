"""Row N4: depth-camera scene clouds (analytic ray cast + on-device subset draw) vs the oracle restatement."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def problem(B, kinds, seed):
    from mpinets_amd.geometry import TorchCuboids, TorchCylinders
    from mpinets_amd.scenes import make_scenes, random_configurations

    scn = make_scenes(B, seed, kinds, 16, 8)
    t = {k: torch.from_numpy(v).to(dev()) for k, v in scn.items()}
    cub = TorchCuboids(t["cuboid_centers"], t["cuboid_dims"], t["cuboid_quats"])
    cyl = TorchCylinders(t["cylinder_centers"], t["cylinder_radii"], t["cylinder_heights"], t["cylinder_quats"])
    q = torch.from_numpy(random_configurations(B, seed + 1)).to(dev())
    return scn, t, cub, cyl, q


def test_depth_image_and_cloud_match_oracle(oracle):
    from mpinets_amd.depth import DepthCamera, camera_pose
    from mpinets_amd.robot import FrankaCollisionSampler

    B = 3
    kinds = ("tabletop", "cubby", "tabletop")
    scn, t, cub, cyl, q = problem(B, ("tabletop", "cubby"), 2)
    kinds = ("tabletop", "cubby", "tabletop")
    cam = DepthCamera(160, 120)  # small image: the scalar oracle walks every pixel x primitive
    poses = torch.from_numpy(np.stack([camera_pose(k) for k in kinds])).to(dev())
    cs = FrankaCollisionSampler(dev(), with_base_link=True)
    depth = cam.render(poses, cub, cyl, q=q, collision_sampler=cs)
    sc = cs.sphere_centers(q).cpu().numpy()
    ref = oracle.depth_render(poses.cpu().numpy(), cam.intrinsics, 160, 120,
                              (scn["cuboid_centers"], scn["cuboid_dims"], scn["cuboid_quats"]),
                              (scn["cylinder_centers"], scn["cylinder_radii"], scn["cylinder_heights"], scn["cylinder_quats"]),
                              sc, cs.radii.cpu().numpy(), cam.far_clip)
    got = depth.reshape(B, -1).cpu().numpy()
    valid = ref >= 0
    assert 0.05 < valid.mean() < 0.98  # the cameras see the scene, and not only the scene
    assert ((got >= 0) != valid).mean() < 1e-4  # silhouette pixels may flip by rounding
    both = (got >= 0) & valid
    np.testing.assert_allclose(got[both], ref[both], rtol=0, atol=5e-6)
    # without the robot more pixels are valid (its spheres hide / remove some)
    no_robot = cam.render(poses, cub, cyl)
    assert (no_robot >= 0).sum() > (depth >= 0).sum()
    # subset draw: same keys, same order as the oracle's sort
    n_out = 500
    pts = cam.sample_cloud(depth, poses, n_out, seed=77)
    opts, ocount = oracle.depth_select(got, poses.cpu().numpy(), cam.intrinsics, 160, 120, n_out, 77)
    np.testing.assert_array_equal(cam.last_counts.cpu().numpy(), ocount)
    np.testing.assert_allclose(pts.cpu().numpy(), opts, rtol=0, atol=2e-6)
    # every point lies on an obstacle surface of its own scene (engine SDF == 0) and is visible, i.e. in front
    sd = torch.minimum(cub.sdf(pts), cyl.sdf(pts)).abs()
    assert sd.max().item() < 2e-5
    p2 = cam.sample_cloud(depth, poses, n_out, seed=78).cpu().numpy()
    assert not np.array_equal(p2, pts.cpu().numpy())


def test_subset_is_uniform_without_replacement_and_errors():
    from mpinets_amd.depth import DepthCamera

    cam = DepthCamera(64, 48)
    B, HW = 2, 64 * 48
    poses = torch.eye(4, device=dev()).repeat(B, 1, 1)
    depth = torch.full((B, 48, 64), -1.0, device=dev())
    depth[0, :, :32] = 1.0 + torch.arange(32, device=dev()) * 0.01  # a distinct depth per column: points identify pixels
    depth[1, 10:40, 5:60] = 2.0
    n_out = 1000
    seen = np.zeros((48, 32), np.int64)
    for seed in range(40):
        pts = cam.sample_cloud(depth, poses, n_out, seed=seed)
        p = pts[0].cpu().numpy()
        # invert the pinhole: pixel from the point (camera at the origin looking along -z)
        u = np.rint(p[:, 0] / -p[:, 2] * cam.fx + cam.cx - 0.5).astype(int)
        v = np.rint(-p[:, 1] / -p[:, 2] * cam.fy + cam.cy - 0.5).astype(int)
        assert len(set(zip(u.tolist(), v.tolist()))) == n_out  # without replacement
        assert u.min() >= 0 and u.max() < 32 and v.min() >= 0 and v.max() < 48  # only valid pixels
        seen[v, u] += 1
    freq = seen / 40.0  # every pixel is chosen with probability n_out / 1536
    assert abs(freq.mean() - n_out / 1536) < 1e-9 and freq.std() < 0.12
    with pytest.raises(ValueError):
        cam.sample_cloud(depth, poses, 1600, seed=0)  # image 0 has only 1536 valid pixels


def test_depth_clouds_feed_the_policy():
    from mpinets_amd.depth import depth_point_clouds
    from mpinets_amd.model import MotionPolicyNetwork
    from mpinets_amd.scenes import make_problem_batch

    B = 4
    prob = make_problem_batch(B, seed=3, device=dev(), kinds=("tabletop", "cubby"), M1=16, M2=8)
    q0 = (prob["q_norm"] + 1) / 2 * 0  # any configuration: neutral-ish robot from the slab's own q is fine
    from mpinets_amd.utils import unnormalize_franka_joints

    q0 = unnormalize_franka_joints(prob["q_norm"])
    kinds = ["tabletop", "cubby", "tabletop", "cubby"]
    depth_point_clouds(prob, q0, kinds, 4096, seed=5, out=prob["xyz"][:, 2048:6144])
    torch.manual_seed(0)
    mdl = MotionPolicyNetwork().to(dev()).eval()
    with torch.no_grad():
        dq = mdl(prob["xyz"], prob["q_norm"])
    assert dq.shape == (B, 7) and torch.isfinite(dq).all()


def test_full_resolution_draw_matches_oracle(oracle):
    """640 x 480 images, the reference's 4096-point draw (run_inference.py:52-54, 78-85): same subset, same order."""
    from mpinets_amd.depth import DepthCamera, camera_pose

    B = 2
    scn, t, cub, cyl, q = problem(B, ("tabletop", "dresser"), 9)
    cam = DepthCamera()
    poses = torch.from_numpy(np.stack([camera_pose("tabletop"), camera_pose("dresser")])).to(dev())
    depth = cam.render(poses, cub, cyl, q=q)
    pts = cam.sample_cloud(depth, poses, 4096, seed=2024)
    opts, ocount = oracle.depth_select(depth.reshape(B, -1).cpu().numpy(), poses.cpu().numpy(), cam.intrinsics, 640, 480,
                                       4096, 2024)
    np.testing.assert_array_equal(cam.last_counts.cpu().numpy(), ocount)
    assert ocount.min() >= 4096
    np.testing.assert_allclose(pts.cpu().numpy(), opts, rtol=0, atol=2e-6)
    assert torch.minimum(cub.sdf(pts), cyl.sdf(pts)).abs().max().item() < 2e-5


def test_render_is_the_first_hit_of_the_reference_pinned_sdf():
    """An oracle-free statement about the ray cast: with the SDF classes whose values are pinned to the reference's
    own ``geometry.py`` (tests/golden), every valid pixel's world point is ON a primitive (|sdf| ~ 0), every point of
    the ray in front of it is OUTSIDE all primitives (sdf > 0: the hit is the first one), and a pixel marked empty has
    no primitive anywhere on its ray up to the far clip (sampled)."""
    from mpinets_amd.depth import DepthCamera, camera_pose

    B, W, H = 2, 96, 72
    kinds = ("tabletop", "cubby")
    scn, t, cub, cyl, q = problem(B, kinds, 5)
    cam = DepthCamera(W, H)
    poses = torch.from_numpy(np.stack([camera_pose(k) for k in kinds])).to(dev())
    depth = cam.render(poses, cub, cyl).reshape(B, H * W)  # no robot: every miss is a true miss
    # ray of every pixel: world = origin + s * dir, OpenGL camera axes (x right, y up, looking along -z), pixel centres
    v, u = torch.meshgrid(torch.arange(H, device=dev()), torch.arange(W, device=dev()), indexing="ij")
    dc = torch.stack([(u + 0.5 - cam.cx) / cam.fx, -(v + 0.5 - cam.cy) / cam.fy, -torch.ones_like(u, dtype=torch.float32)], -1)
    dc = (dc / dc.norm(dim=-1, keepdim=True)).reshape(1, H * W, 3).float()
    R, o = poses[:, :3, :3], poses[:, None, :3, 3]
    dw = torch.einsum("bij,bpj->bpi", R, dc.expand(B, -1, -1))
    sdf = lambda p: torch.minimum(cub.sdf(p.contiguous()), cyl.sdf(p.contiguous()))
    valid = depth >= 0
    assert 0.05 < valid.float().mean().item() < 0.98
    hit = o + depth.clamp(min=0)[..., None] * dw
    assert sdf(hit)[valid].abs().max().item() < 5e-5  # on a surface
    for frac in (0.1, 0.3, 0.5, 0.7, 0.9, 0.97):  # nothing in front of the hit
        s = sdf(o + (frac * depth.clamp(min=0))[..., None] * dw)
        assert (s[valid] > -1e-5).all(), frac
    miss = ~valid
    for dist in np.linspace(0.2, cam.far_clip, 60):  # an empty pixel's ray never enters a primitive (sampled every 17 cm:
        s = sdf(o + float(dist) * dw)                 # the thinnest plates are 1 cm, so this is a one-sided check)
        assert (s[miss] > -5e-3).all(), dist
