"""CPU: the oracle's restatement of model.py vs vectors produced by RUNNING the reference's model.py
(tests/golden/gen_model_golden.py): heads, fc layer, slab split, full forward, rollout loop, validation reduce.

Tolerance 1e-6 (the oracle accumulates in float64 and stores float32; the golden is torch's fp32 CPU arithmetic).
What this pins: SURVEY.md rows a6, a13 (heads / composition), a14, a15.  What it cannot pin: the indices and the FK that
the generator's stubs take from this same oracle (pointnet2_ops / robofin are absent from the container)."""
import numpy as np
import pytest

import seeded_weights

TOL = 1e-6
NR = 2048
SCENE_KEYS = ("cuboid_centers", "cuboid_dims", "cuboid_quats", "cylinder_centers", "cylinder_radii", "cylinder_heights",
              "cylinder_quats")


def golden_state_dict(g):
    shapes = {str(n): tuple(int(x) for x in s[:np.count_nonzero(s)]) for n, s in zip(g["param_names"], g["param_shapes"])}
    sd = seeded_weights.seeded_state_dict(shapes, seed=0)
    assert seeded_weights.digest(sd) == str(g["param_sha256"]), "regenerated weights differ from the ones the golden was made with"
    return sd


@pytest.fixture(scope="module")
def sd(model_golden):
    return golden_state_dict(model_golden)


def _limits():
    from mpinets_amd import franka_tables as ft

    return ft.JOINT_LIMITS_REAL


def _tables():
    from mpinets_amd import franka_tables as ft

    return ft.link_point_table(4096, True)


def test_parameter_names_and_shapes_are_the_reference_models(model_golden, sd):
    """The generator loaded these names/shapes into the reference's TrainingMotionPolicyNetwork with strict=True."""
    assert int(model_golden["num_params"]) == sum(v.size for v in sd.values()) == 19068103
    assert sd["point_cloud_encoder.SA_modules.0.mlps.0.0.weight"].shape == (64, 4, 1, 1)
    assert sd["point_cloud_encoder.SA_modules.1.mlps.0.0.weight"].shape == (128, 67, 1, 1)
    assert sd["point_cloud_encoder.SA_modules.2.mlps.0.0.weight"].shape == (512, 259, 1, 1)
    assert sd["decoder.0.weight"].shape == (512, 2048 + 64)


def test_heads_fc_layer_and_slab_split(oracle, model_golden, sd):
    g = model_golden
    x = g["h_q"]
    for k in (0, 2, 4, 6, 8):
        x = oracle._linear(x, sd[f"feature_encoder.{k}.weight"], sd[f"feature_encoder.{k}.bias"])
        x = oracle._leaky(x) if k != 8 else x
    np.testing.assert_allclose(x, g["h_feature"], rtol=0, atol=TOL)
    x = g["h_dec_in"]
    for k in (0, 2, 4, 6):
        x = oracle._linear(x, sd[f"decoder.{k}.weight"], sd[f"decoder.{k}.bias"])
        x = oracle._leaky(x) if k != 6 else x
    np.testing.assert_allclose(x, g["h_dec_out"], rtol=0, atol=TOL)
    x = oracle.fc_layer(sd, g["h_fc_in"])
    assert np.abs(g["h_fc_out"]).max() > 0.5
    np.testing.assert_allclose(x, g["h_fc_out"], rtol=0, atol=2 * TOL)
    xyz, feat = oracle.break_up_pc(g["h_pc"])
    np.testing.assert_array_equal(xyz, g["h_pc_xyz"])
    np.testing.assert_array_equal(feat, g["h_pc_features"])


def test_forward_and_every_module_output(oracle, model_golden, sd):
    g = model_golden
    dq, aux = oracle.policy_forward(sd, g["f_xyz"], g["f_q"])
    np.testing.assert_array_equal(aux["sa1"]["fps_idx"], g["f_fps1"])
    np.testing.assert_array_equal(aux["sa2"]["fps_idx"], g["f_fps2"])
    np.testing.assert_array_equal(aux["xyz1"], g["f_xyz1"])
    np.testing.assert_array_equal(aux["xyz2"], g["f_xyz2"])
    for mine, ref in ((aux["f1"], g["f_feat1"]), (aux["f2"], g["f_feat2"]), (aux["f3"], g["f_feat3"]),
                      (aux["encoding"], g["f_encoding"]), (dq, g["f_out"])):
        assert mine.shape == ref.shape
        np.testing.assert_allclose(mine, ref, rtol=0, atol=2 * TOL)
    assert np.abs(g["f_out"]).max() > 0.05


@pytest.mark.parametrize("tag,unnorm", [("n", False), ("u", True)])
def test_rollout_loop(oracle, model_golden, sd, tag, unnorm):
    g = model_golden
    pts, link = _tables()
    slab = g["f_xyz"][:2].copy()
    sampler = lambda qu, i: oracle.transform_table(oracle.franka_fk(qu), pts, link, g["r_subsets"][i])
    traj = oracle.rollout(sd, slab, g["f_q"][:2], 5, sampler, _limits(), unnormalize_out=unnorm)
    np.testing.assert_allclose(np.stack(traj), g[f"r_traj_{tag}"], rtol=0, atol=5 * TOL)
    np.testing.assert_allclose(slab[:, :NR, :3], g[f"r_robot_{tag}"], rtol=0, atol=5 * TOL)
    np.testing.assert_array_equal(slab[:, NR:], g["f_xyz"][:2, NR:])  # only the robot rows are rewritten
    np.testing.assert_array_equal(slab[:, :NR, 3], g["f_xyz"][:2, :NR, 3])
    # the clamp is exercised: some joints sit on +-1, and they are exactly +-1 on both sides
    on_limit = np.abs(g["r_traj_n"]) == 1
    assert on_limit.any()
    if not unnorm:
        np.testing.assert_array_equal(np.abs(np.stack(traj)) == 1, on_limit)


def test_rollout_single_trajectory_form(oracle, model_golden, sd):
    g = model_golden
    pts, link = _tables()
    slab = g["f_xyz"][2:3].copy()
    sampler = lambda qu, i: oracle.transform_table(oracle.franka_fk(qu), pts, link, g["r1_subsets"][i])
    traj = oracle.rollout(sd, slab, g["f_q"][2:3], 2, sampler, _limits())
    np.testing.assert_allclose(np.stack(traj), g["r1_traj"], rtol=0, atol=5 * TOL)
    np.testing.assert_allclose(slab[0, :NR, :3], g["r1_robot"], rtol=0, atol=5 * TOL)


def _prims(g, prefix):
    return tuple(g[prefix + k] for k in SCENE_KEYS[:3]), tuple(g[prefix + k] for k in SCENE_KEYS[3:])


@pytest.mark.parametrize("prefix,traj_key", [("c_", "c_traj"), ("v_", "v_traj")])
def test_validation_collision_reduce(oracle, model_golden, prefix, traj_key):
    """model.py:293-314 executed by the reference on given 70-waypoint rollouts vs the oracle's fused restatement."""
    from mpinets_amd import franka_tables as ft

    g = model_golden
    c, r, l, _ = ft.collision_sphere_table(False)
    cub, cyl = _prims(g, prefix)
    res = oracle.validation_reduce(g[traj_key], g[prefix + "target_position"], (c, r, l), cub, cyl)
    np.testing.assert_array_equal(res["has_collision"], g[prefix + "flags"])
    np.testing.assert_allclose(res["margin"], g[prefix + "margin"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(res["position_error"], g[prefix + "errors"], rtol=0, atol=TOL)
    assert abs(res["avg_collision_rate"] - float(g[prefix + "rate"])) < 1e-7
    assert abs(res["avg_target_error"] - float(g[prefix + "target_error"])) < TOL
    assert 0 < g[prefix + "flags"].sum() < len(g[prefix + "flags"])  # both outcomes are present


def test_validation_closed_loop_steps_teacher_forced(oracle, model_golden, sd):
    """Steps 0, 1, 34 and 68 of the reference's own 69-step validation rollout, each from the reference's state."""
    g = model_golden
    pts, link = _tables()
    lim = _limits()
    traj = g["v_traj"]  # [3,70,7] joint angles
    for i in (0, 1, 34, 68):
        slab = g["f_xyz"].copy()
        if i > 0:
            slab[:, :NR, :3] = oracle.transform_table(oracle.franka_fk(traj[:, i]), pts, link, g["v_subsets"][i - 1])
        qn = oracle.normalize(traj[:, i], lim) if i > 0 else g["f_q"]
        dq, _ = oracle.policy_forward(sd, slab, qn)
        nxt = oracle.unnormalize(np.clip(qn + dq, -1, 1).astype(np.float32), lim)
        np.testing.assert_allclose(nxt, traj[:, i + 1], rtol=0, atol=5 * TOL)


def test_training_step_loss_and_gradients(oracle, model_golden, sd):
    """model.py:185-240 + backward, run by the reference (losses 1 : 5 like jobconfig.yaml): the oracle's float64 autograd
    restatement gives the same loss and the same gradients for parameters of every part of the network."""
    import torch

    from mpinets_amd import franka_tables as ft

    g = model_golden
    leaf = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in sd.items()}
    q = torch.tensor(g["f_q"][:2], dtype=torch.float64)
    y = torch.clamp(q + oracle.policy_forward_torch(leaf, g["f_xyz"][:2], q), -1, 1)
    lim = torch.tensor(ft.JOINT_LIMITS_REAL, dtype=torch.float64)
    unnorm = lambda x: (x + 1) * (lim[:, 1] - lim[:, 0]) / 2 + lim[:, 0]
    pts, link = ft.link_point_table(4096, with_base_link=False)
    cloud = oracle.robot_cloud_torch(unnorm(y), pts, link, g["t_fixed_subset"])
    target = oracle.robot_cloud_torch(unnorm(torch.tensor(g["t_supervision"], dtype=torch.float64)), pts, link, g["t_fixed_subset"])
    t64 = lambda k: torch.tensor(g["v_" + k][:2], dtype=torch.float64)
    cf = torch.tensor(oracle.inv_frames_4x4(g["v_cuboid_centers"][:2], g["v_cuboid_quats"][:2]), dtype=torch.float64)
    yf = torch.tensor(oracle.inv_frames_4x4(g["v_cylinder_centers"][:2], g["v_cylinder_quats"][:2]), dtype=torch.float64)
    coll = oracle.collision_loss_torch(cloud, cf, t64("cuboid_dims"), yf, t64("cylinder_radii")[..., 0], t64("cylinder_heights")[..., 0])
    loss = oracle.point_match_loss_torch(cloud, target) + 5.0 * coll
    loss.backward()
    assert abs(loss.item() - float(g["t_loss"])) < 2e-6
    for k in g:
        if k.startswith("t_grad."):
            ref, mine = g[k], leaf[k[7:]].grad.numpy()
            assert np.abs(mine - ref).max() <= 2e-4 * np.abs(ref).max() + 1e-9, (k, np.abs(mine - ref).max(), np.abs(ref).max())
