"""Row N1: HIP losses + analytic gradients vs vectors made by the reference's loss.py, and vs the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ORDER = ("cuboid_centers", "cuboid_dims", "cuboid_quats", "cylinder_centers", "cylinder_radii", "cylinder_heights",
         "cylinder_quats")


def dev():
    return torch.device("cuda:0")


def scene(g, prefix):
    return [torch.tensor(g[prefix + k]).to(dev()) for k in ORDER]


def test_collision_loss_and_gradient_match_reference(loss_golden):
    from mpinets_amd.loss import collision_loss

    g = loss_golden
    pc = torch.tensor(g["c_points"], device=dev(), requires_grad=True)
    loss = collision_loss(pc, *scene(g, "c_"))
    loss.backward()
    assert abs(loss.item() - float(g["c_loss"])) < 1e-6  # tolerance: 1e-5 stated by the north star, 1e-6 held
    scale = np.abs(g["c_grad"]).max()
    np.testing.assert_allclose(pc.grad.cpu().numpy(), g["c_grad"], atol=1e-5 * scale, rtol=1e-4)
    # forward only (no gradient requested) gives the same value and allocates no gradient
    with torch.no_grad():
        assert collision_loss(pc, *scene(g, "c_")).item() == loss.item()


def test_point_match_loss_and_gradient_match_reference(loss_golden):
    from mpinets_amd.loss import point_match_loss

    g = loss_golden
    a = torch.tensor(g["p_input"], device=dev(), requires_grad=True)
    loss = point_match_loss(a, torch.tensor(g["p_target"], device=dev()))
    (3.0 * loss).backward()
    assert abs(loss.item() - float(g["p_loss"])) < 1e-7
    np.testing.assert_allclose(a.grad.cpu().numpy() / 3.0, g["p_grad"], atol=1e-10, rtol=1e-5)
    assert (a.grad[0, :10] == 0).all()  # exact matches: sign(0) = 0 like torch


def test_loss_container_matches_reference(loss_golden):
    from mpinets_amd.loss import CollisionAndBCLossContainer
    from mpinets_amd.robot import FrankaSampler

    g = loss_golden
    box = CollisionAndBCLossContainer()
    box.fk_sampler = FrankaSampler(dev(), num_fixed_points=1024, use_cache=True, with_base_link=False)
    box.fk_sampler._fixed = torch.tensor(g["subset"], device=dev())  # the subset the vectors were made with
    x = torch.tensor(g["k_input"], device=dev(), requires_grad=True)
    coll, pm = box(x, *scene(g, "k_"), torch.tensor(g["k_target"], device=dev()))
    gc, = torch.autograd.grad(coll, x, retain_graph=True)
    gp, = torch.autograd.grad(pm, x)
    assert abs(coll.item() - float(g["k_collision_loss"])) < 1e-6
    assert abs(pm.item() - float(g["k_point_match_loss"])) < 1e-6
    np.testing.assert_allclose(gc.cpu().numpy(), g["k_grad_collision"], atol=1e-5 * np.abs(g["k_grad_collision"]).max())
    np.testing.assert_allclose(gp.cpu().numpy(), g["k_grad_point_match"],
                               atol=1e-5 * np.abs(g["k_grad_point_match"]).max())


def test_training_step_shape_batch_vs_oracle(oracle):
    """Bigger batch, tabletop scenes, clamp in front like model.py:202: HIP vs the float64 oracle autograd."""
    from mpinets_amd import franka_tables as ft
    from mpinets_amd.loss import CollisionAndBCLossContainer
    from mpinets_amd.scenes import make_scenes, random_configurations
    from mpinets_amd.utils import normalize_franka_joints

    B = 96
    scn = make_scenes(B, 3, ("tabletop",), 12, 8)
    rng = np.random.default_rng(2)
    qn = normalize_franka_joints(random_configurations(B, 4).astype(np.float64)).astype(np.float32)
    dq = rng.normal(scale=0.3, size=qn.shape).astype(np.float32)
    tn = np.clip(qn + rng.normal(scale=0.1, size=qn.shape), -1, 1).astype(np.float32)
    names = ("cuboid_centers", "cuboid_dims", "cuboid_quats", "cylinder_centers", "cylinder_radii",
             "cylinder_heights", "cylinder_quats")
    box = CollisionAndBCLossContainer()
    d = torch.tensor(dq, device=dev(), requires_grad=True)
    y = torch.clamp(torch.tensor(qn, device=dev()) + d, min=-1, max=1)
    coll, pm = box(y, *(torch.tensor(scn[k]).to(dev()) for k in names), torch.tensor(tn, device=dev()))
    (5.0 * coll + pm).backward()
    first = d.grad.clone()
    # determinism: fixed-order reductions, no atomics
    d.grad = None
    y2 = torch.clamp(torch.tensor(qn, device=dev()) + d, min=-1, max=1)
    c2, p2 = box(y2, *(torch.tensor(scn[k]).to(dev()) for k in names), torch.tensor(tn, device=dev()))
    (5.0 * c2 + p2).backward()
    assert torch.equal(first, d.grad) and c2.item() == coll.item() and p2.item() == pm.item()

    # oracle, float64
    pts, link = ft.link_point_table(4096, with_base_link=False)
    sub = box.fk_sampler._fixed.cpu().numpy()
    lim = torch.tensor(ft.JOINT_LIMITS_REAL, dtype=torch.float64)
    unnorm = lambda x: (x + 1) * (lim[:, 1] - lim[:, 0]) / 2 + lim[:, 0]
    dd = torch.tensor(dq, dtype=torch.float64, requires_grad=True)
    yy = torch.clamp(torch.tensor(qn, dtype=torch.float64) + dd, min=-1, max=1)
    cloud = oracle.robot_cloud_torch(unnorm(yy), pts, link, sub)
    target = oracle.robot_cloud_torch(unnorm(torch.tensor(tn, dtype=torch.float64)), pts, link, sub)
    f64 = lambda k: torch.tensor(scn[k], dtype=torch.float64)
    cf = torch.tensor(oracle.inv_frames_4x4(scn["cuboid_centers"], scn["cuboid_quats"]), dtype=torch.float64)
    yf = torch.tensor(oracle.inv_frames_4x4(scn["cylinder_centers"], scn["cylinder_quats"]), dtype=torch.float64)
    oc = oracle.collision_loss_torch(cloud, cf, f64("cuboid_dims"), yf, f64("cylinder_radii")[..., 0],
                                     f64("cylinder_heights")[..., 0])
    op = oracle.point_match_loss_torch(cloud, target)
    (5.0 * oc + op).backward()
    assert abs(coll.item() - oc.item()) < 1e-6 and abs(pm.item() - op.item()) < 1e-6
    ref = dd.grad.numpy()
    assert np.abs(ref).max() > 0
    np.testing.assert_allclose(first.cpu().numpy(), ref, atol=2e-5 * np.abs(ref).max())


def test_empty_scene_and_cpu_tensor_errors():
    from mpinets_amd import _lib
    from mpinets_amd.loss import collision_loss, point_match_loss

    z = lambda *s: torch.zeros(*s, device=dev())
    pc = torch.rand(2, 64, 3, device=dev(), requires_grad=True)
    quat = torch.tensor([1.0, 0, 0, 0], device=dev()).expand(2, 1, 4).contiguous()
    loss = collision_loss(pc, z(2, 1, 3), z(2, 1, 3), quat, z(2, 1, 3), z(2, 1, 1), z(2, 1, 1), quat)
    loss.backward()
    assert loss.item() == 0 and (pc.grad == 0).all()  # zero-volume primitives: sdf = +inf, no loss, no gradient
    with pytest.raises(_lib.MpxError):
        point_match_loss(torch.rand(2, 4, 3), torch.rand(2, 4, 3))
