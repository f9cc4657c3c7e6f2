"""Row N1, CPU: the oracle's differentiable restatement of the losses vs vectors made by the reference's loss.py."""
import numpy as np
import torch

ORDER = ("cuboid_centers", "cuboid_dims", "cuboid_quats", "cylinder_centers", "cylinder_radii", "cylinder_heights",
         "cylinder_quats")


def scene_tensors(oracle, g, prefix, dtype=torch.float64):
    cf = torch.tensor(oracle.inv_frames_4x4(g[prefix + "cuboid_centers"], g[prefix + "cuboid_quats"]), dtype=dtype)
    yf = torch.tensor(oracle.inv_frames_4x4(g[prefix + "cylinder_centers"], g[prefix + "cylinder_quats"]), dtype=dtype)
    t = lambda k: torch.tensor(g[prefix + k], dtype=dtype)
    return cf, t("cuboid_dims"), yf, t("cylinder_radii")[..., 0], t("cylinder_heights")[..., 0]


def test_collision_loss_restatement_matches_reference(oracle, loss_golden):
    g = loss_golden
    pc = torch.tensor(g["c_points"], dtype=torch.float64, requires_grad=True)
    loss = oracle.collision_loss_torch(pc, *scene_tensors(oracle, g, "c_"))
    loss.backward()
    assert abs(loss.item() - float(g["c_loss"])) < 2e-6
    np.testing.assert_allclose(pc.grad.numpy(), g["c_grad"], atol=2e-6 * np.abs(g["c_grad"]).max() + 1e-9, rtol=2e-3)


def test_point_match_restatement_matches_reference(oracle, loss_golden):
    g = loss_golden
    a = torch.tensor(g["p_input"], dtype=torch.float64, requires_grad=True)
    loss = oracle.point_match_loss_torch(a, torch.tensor(g["p_target"], dtype=torch.float64))
    loss.backward()
    assert abs(loss.item() - float(g["p_loss"])) < 1e-7
    np.testing.assert_allclose(a.grad.numpy(), g["p_grad"], atol=1e-9, rtol=1e-5)


def test_container_restatement_matches_reference(oracle, loss_golden):
    from mpinets_amd import franka_tables as ft

    g = loss_golden
    pts, link = ft.link_point_table(4096, with_base_link=False)
    lim = torch.tensor(ft.JOINT_LIMITS_REAL, dtype=torch.float64)
    unnorm = lambda x: (x + 1) * (lim[:, 1] - lim[:, 0]) / 2 + lim[:, 0]
    x = torch.tensor(g["k_input"], dtype=torch.float64, requires_grad=True)
    cloud = oracle.robot_cloud_torch(unnorm(x), pts, link, g["subset"])
    target = oracle.robot_cloud_torch(unnorm(torch.tensor(g["k_target"], dtype=torch.float64)), pts, link, g["subset"])
    coll = oracle.collision_loss_torch(cloud, *scene_tensors(oracle, g, "k_"))
    pm = oracle.point_match_loss_torch(cloud, target)
    gc, = torch.autograd.grad(coll, x, retain_graph=True)
    gp, = torch.autograd.grad(pm, x)
    assert abs(coll.item() - float(g["k_collision_loss"])) < 2e-6
    assert abs(pm.item() - float(g["k_point_match_loss"])) < 1e-6
    np.testing.assert_allclose(gc.numpy(), g["k_grad_collision"], atol=2e-5 * np.abs(g["k_grad_collision"]).max())
    np.testing.assert_allclose(gp.numpy(), g["k_grad_point_match"], atol=1e-5 * np.abs(g["k_grad_point_match"]).max())


def test_fk_torch_restatement_equals_c_oracle(oracle):
    q = np.random.default_rng(1).uniform(-2, 2, (16, 7)).astype(np.float32)
    R, t = oracle.fk_frames_torch(torch.tensor(q, dtype=torch.float64))
    T = oracle.franka_fk(q)
    np.testing.assert_allclose(R.numpy().reshape(16, 15, 9), T[..., :9], atol=2e-6)
    np.testing.assert_allclose(t.numpy(), T[..., 9:], atol=2e-6)


def test_batch_config_restatement_properties(oracle):
    """Row N2 oracle: no noise -> the stored waypoint; noise is clamped to the limits; supervision = next waypoint
    (last one re-used); target = FK of the LAST waypoint (data_loader.py:155-185, 403-417)."""
    from mpinets_amd import franka_tables as ft

    rng = np.random.default_rng(3)
    lim = ft.JOINT_LIMITS_REAL
    traj = (lim[:, 0] + rng.random((4, 6, 7)) * (lim[:, 1] - lim[:, 0])).astype(np.float32)
    traj[2, 1] = lim[:, 1]  # on the upper limits: noise can only push it inside or get clamped
    o = oracle.batch_configs(traj, [0, 2, 3], [0, 1, 5], lim, 0.0, 1)
    np.testing.assert_array_equal(o["q"], traj[[0, 2, 3], [0, 1, 5]])
    np.testing.assert_allclose(o["supervision"][0], oracle.normalize(traj[0, 1][None], lim)[0], atol=1e-6)
    np.testing.assert_allclose(o["supervision"][2], oracle.normalize(traj[3, 5][None], lim)[0], atol=1e-6)
    np.testing.assert_allclose(o["target_pose"][1], oracle.frames_to_4x4(oracle.franka_fk(traj[2, 5][None])[:, 14])[0])
    n = oracle.batch_configs(traj, [2] * 50, [1] * 50, lim, 0.05, 7)
    assert (n["q"] <= lim[:, 1] + 1e-6).all() and (n["q"] >= lim[:, 0] - 1e-6).all() and np.abs(n["configuration"]).max() <= 1 + 1e-6
    assert (n["q"] < lim[:, 1]).mean() > 0.3 and len(np.unique(n["q"][:, 0])) > 10  # different noise per sample
