"""BASELINE-size checks (8192 mixed environments per GPU) through properties that do not need the scalar oracle:
index validity / order / radius of the neighbour search, farthest-point invariants, fused vs unfused collision
sweep, FK cloud vs slab, run-to-run determinism; a few environments are also compared with the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
B = 8192


def dev():
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def problem():
    from mpinets_amd.scenes import make_problem_batch

    return make_problem_batch(B, seed=77, device=dev(), kinds=("tabletop", "cubby", "dresser"), M1=40, M2=16,
                              scene_pool=512, device_clouds=True)


@pytest.fixture(scope="module")
def forward(problem):
    from mpinets_amd.model import MotionPolicyNetwork

    torch.manual_seed(0)
    mdl = MotionPolicyNetwork().to(dev()).eval()
    aux = {}
    with torch.no_grad():
        dq = mdl(problem["xyz"], problem["q_norm"], aux=aux)
    return mdl, dq, aux


def test_fps_invariants_at_full_size(problem, forward, oracle):
    _, _, aux = forward
    for key, n in (("fps_idx1", 6272), ("fps_idx2", 512)):
        idx = aux[key].long()
        assert idx.min() >= 0 and idx.max() < n and (idx[:, 0] == 0).all()
        assert (torch.sort(idx, dim=1).values.diff(dim=1) > 0).all()  # no point is picked twice
    # the sequence of minimum distances to the already-picked set never increases (farthest-point property)
    xyz = problem["xyz"][:256, :, :3]
    picks = torch.gather(xyz, 1, aux["fps_idx1"][:256].long()[:, :, None].expand(-1, -1, 3))
    d = torch.cdist(picks.double(), picks.double())  # [256,512,512]
    tri = torch.tril(torch.ones(512, 512, device=dev(), dtype=torch.bool), diagonal=-1)
    mind = torch.where(tri, d, torch.full_like(d, float("inf"))).min(dim=2).values[:, 1:]  # pick j vs picks < j
    assert (mind[:, 1:] <= mind[:, :-1] + 1e-6).all(), (mind[:, 1:] - mind[:, :-1]).max().item()
    # and a few environments bit-for-bit against the scalar oracle
    sel = [0, 4097, 8191]
    np.testing.assert_array_equal(aux["fps_idx1"][sel].cpu().numpy(), oracle.fps(problem["xyz"][sel].cpu().numpy(), 512))


@pytest.mark.parametrize("stage", [1, 2])
def test_ball_query_properties_at_full_size(problem, forward, oracle, stage):
    _, _, aux = forward
    if stage == 1:
        cloud, centres, r, N = problem["xyz"][:, :, :3], aux["xyz1"], 0.05, 6272
    else:
        cloud, centres, r, N = aux["xyz1"], aux["sa3_in"][:, :, :3], 0.3, 512
    idx, cnt = aux[f"ball_idx{stage}"], aux[f"ball_cnt{stage}"]
    assert idx.min() >= 0 and idx.max() < N and cnt.min() >= 0 and cnt.max() <= 128
    slot = torch.arange(128, device=dev())[None, None, :]
    real = slot < cnt[:, :, None]
    # real slots: strictly increasing point indices; padding: copies of the first hit (or 0 when nothing was hit)
    inc = (idx[:, :, 1:] > idx[:, :, :-1]) | ~real[:, :, 1:]
    assert inc.all()
    first = torch.where(cnt > 0, idx[:, :, 0], torch.zeros_like(idx[:, :, 0]))
    assert (torch.where(real, first[:, :, None].expand_as(idx), idx) == first[:, :, None]).all()
    # every real slot is inside the radius (same fp32 expression as the kernels), checked on a slice of the batch
    for b0 in range(0, B, 2048):
        sl = slice(b0, b0 + 256)
        nb = torch.gather(cloud[sl], 1, idx[sl].long().reshape(256, -1)[:, :, None].expand(-1, -1, 3)).reshape(256, -1, 128, 3)
        d = centres[sl][:, :, None, :] - nb
        d2 = torch.addcmul(torch.addcmul(d[..., 0] * d[..., 0], d[..., 1], d[..., 1]), d[..., 2], d[..., 2])
        assert (d2[real[sl]] < r * r * (1 + 1e-6)).all()
        # the count is the number of points within the radius (capped at nsample)
        dist2 = torch.cdist(centres[sl].double(), cloud[sl].double()) ** 2
        within = (dist2 < r * r).sum(dim=2)
        close = ((dist2 - r * r).abs() < 1e-7).any(dim=2)  # pairs sitting on the sphere: rounding may differ
        ok = (torch.minimum(within, torch.tensor(128, device=dev())) == cnt[sl]) | close
        assert ok.all()
    sel = [1, 5000]
    if stage == 1:
        ref, rcnt = oracle.ball_query(centres[sel].cpu().numpy(), problem["xyz"][sel].cpu().numpy(), r, 128, return_counts=True)
        np.testing.assert_array_equal(idx[sel].cpu().numpy(), ref)
        np.testing.assert_array_equal(cnt[sel].cpu().numpy(), rcnt)


def test_policy_step_is_deterministic_and_finite(problem, forward):
    mdl, dq, aux = forward
    with torch.no_grad():
        dq2 = mdl(problem["xyz"], problem["q_norm"])
    assert torch.equal(dq, dq2) and torch.isfinite(dq).all() and dq.shape == (B, 7)
    assert torch.isfinite(aux["sa3_in"]).all() and (aux["f1"] >= 0).all()  # pooled ReLU outputs


def test_single_call_forward_matches_at_full_size(problem, forward):
    """mpx_policy_forward (one C call, 11.5 GB workspace at this size) == the Python-orchestrated forward, bit for bit."""
    mdl, dq, _ = forward
    with torch.no_grad():
        dq_c = mdl.forward_native(problem["xyz"], problem["q_norm"])
    assert torch.equal(dq_c, dq)


def test_collision_sweep_fused_equals_unfused_at_config4_size(problem):
    """BASELINE config 4: 8192 trajectories x 50 waypoints; fused kernel vs sphere centres + SDF classes + threshold."""
    from mpinets_amd.geometry import TorchCuboids, TorchCylinders
    from mpinets_amd.robot import FrankaCollisionSampler
    from mpinets_amd.scenes import linear_trajectories

    traj = torch.from_numpy(linear_trajectories(B, 50, 3)).to(dev())
    cub = TorchCuboids(problem["cuboid_centers"], problem["cuboid_dims"], problem["cuboid_quats"])
    cyl = TorchCylinders(problem["cylinder_centers"], problem["cylinder_radii"], problem["cylinder_heights"],
                         problem["cylinder_quats"])
    cs = FrankaCollisionSampler(dev(), with_base_link=False)
    fused = cs.check(traj, cub, cyl)
    hit = torch.zeros(B, dtype=torch.bool, device=dev())
    for radius, centres in cs.compute_spheres(traj.reshape(-1, 7)):  # the reference's loop (model.py:300-312)
        c = centres.reshape(B, 50, -1, 3)
        sdf = torch.minimum(cub.sdf_sequence(c), cyl.sdf_sequence(c))
        hit |= (sdf <= radius).reshape(B, -1).any(dim=1)
    assert torch.equal(fused, hit) and 0.02 < fused.float().mean().item() < 0.98


def test_config1_fk_sdf_1024_envs_against_the_oracle(problem, oracle):
    """BASELINE configs[1] as SURVEY 8(d) C2 states it: 1024 environments, one configuration each drawn uniformly inside
    the joint limits, FK + swept-sphere SDF only, outputs has_collision [1024] and min-sdf [1024, 56].  A sample of 32
    environments spread over the batch is compared with the oracle (distances <= 1e-5, flags exact away from the
    radius); the whole batch through size-independent properties: T = 1 equals the [B, 7] form, the flag is
    `any(min_sdf <= radius)`, and rows do not depend on their neighbours in the batch."""
    from mpinets_amd.geometry import TorchCuboids, TorchCylinders
    from mpinets_amd.robot import FrankaCollisionSampler
    from mpinets_amd.scenes import random_configurations

    n = 1024
    q = torch.from_numpy(random_configurations(n, 6)).to(dev())
    prim = lambda sl: (TorchCuboids(problem["cuboid_centers"][sl], problem["cuboid_dims"][sl], problem["cuboid_quats"][sl]),
                       TorchCylinders(problem["cylinder_centers"][sl], problem["cylinder_radii"][sl],
                                      problem["cylinder_heights"][sl], problem["cylinder_quats"][sl]))
    cs = FrankaCollisionSampler(dev(), with_base_link=False)
    cub, cyl = prim(slice(0, n))
    flags, msdf = cs.check(q, cub, cyl, return_sdf=True)
    assert flags.shape == (n,) and msdf.shape == (n, 1, 56) and flags.dtype == torch.bool
    f3, m3 = cs.check(q[:, None, :], cub, cyl, return_sdf=True)
    assert torch.equal(flags, f3) and torch.equal(msdf, m3)
    assert torch.equal(flags, (msdf[:, 0] <= cs.radii[None]).any(dim=1)) and 0.02 < flags.float().mean().item() < 0.98
    sel = np.linspace(0, n - 1, 32).astype(np.int64)
    sel_t = torch.from_numpy(sel).to(dev())
    cub_s, cyl_s = prim(sel_t)
    f_s, m_s = cs.check(q[sel_t], cub_s, cyl_s, return_sdf=True)
    assert torch.equal(f_s, flags[sel_t]) and torch.equal(m_s, msdf[sel_t])  # independent of the batch around a row
    qh = q[sel_t].cpu().numpy()
    centres = oracle.transform_table(oracle.franka_fk(qh), cs.centers.cpu().numpy(), cs.links.cpu().numpy()).reshape(32, 1, 56, 3)
    h = lambda k: problem[k][sel_t].cpu().numpy()
    oflags, omsdf = oracle.collision_flags(
        centres, cs.radii.cpu().numpy(), (h("cuboid_centers"), h("cuboid_dims"), h("cuboid_quats")),
        (h("cylinder_centers"), h("cylinder_radii"), h("cylinder_heights"), h("cylinder_quats")))
    np.testing.assert_allclose(m_s.cpu().numpy(), omsdf, rtol=0, atol=1e-5)
    border = np.abs(omsdf - cs.radii.cpu().numpy()[None, None]).min(axis=(1, 2)) < 1e-5
    np.testing.assert_array_equal(f_s.cpu().numpy()[~border], oflags[~border])


def test_rollout_step_keeps_the_slab_consistent(problem):
    from mpinets_amd.model import MotionPolicyNetwork
    from mpinets_amd.rollout import RolloutEngine

    torch.manual_seed(0)
    mdl = MotionPolicyNetwork().to(dev()).eval()
    prob = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in problem.items()}
    scene_before = prob["xyz"][:, 2048:].clone()
    eng = RolloutEngine(mdl, prob)
    q = eng.step()
    assert torch.equal(prob["xyz"][:, 2048:], scene_before)  # only the robot rows are rewritten (model.py:180-181)
    again = torch.empty((B, 2048, 3), device=dev())
    eng.sampler.sample_into(q, again, eng.subset)
    assert torch.equal(prob["xyz"][:, :2048, :3], again) and (prob["xyz"][:, :2048, 3] == 0).all()
    lim = eng.limits
    assert (q >= lim[:, 0] - 1e-6).all() and (q <= lim[:, 1] + 1e-6).all()


def test_batches_beyond_one_launch_run_in_slabs():
    """B >= 16 384 environments per GPU used to fail (a GEMM over B * 512 rows needs more than 65 535
    row blocks of gridDim.y).  The launchers now walk row / batch slabs: B = 20 000 through the single-call C forward
    (mpx_policy_forward) and through the Python path; rows are independent, so the first and the last 1 040
    environments equal their own 1 040-environment forward bit for bit (same launch-shape regime)."""
    from mpinets_amd.model import MotionPolicyNetwork
    from mpinets_amd.scenes import make_problem_batch

    dev = torch.device("cuda:0")
    torch.manual_seed(2)
    mdl = MotionPolicyNetwork().to(dev).eval()
    B, n = 20000, 1040
    prob = make_problem_batch(B, seed=77, device=dev, kinds=("tabletop", "cubby", "dresser"), M1=40, M2=16, scene_pool=256,
                              device_clouds=True)
    with torch.no_grad():
        dq_native = mdl.forward_native(prob["xyz"], prob["q_norm"]).clone()
        dq_py = mdl(prob["xyz"], prob["q_norm"]).clone()
        assert torch.equal(dq_native, dq_py)
        for sl in (slice(0, n), slice(B - n, B)):
            part = mdl(prob["xyz"][sl].contiguous(), prob["q_norm"][sl].contiguous())
            assert torch.equal(part, dq_py[sl]), sl
    assert torch.isfinite(dq_py).all() and dq_py.abs().max() > 0
    # the batched launchers with a batch axis on gridDim.y: more than 65 535 environments
    from mpinets_amd.pointnet2 import ball_query
    from mpinets_amd.scenes import sample_scene_clouds

    Bq = 65540
    xyz = torch.rand((Bq, 96, 3), device=dev)
    q = xyz[:, :5].contiguous()
    idx, cnt = ball_query(0.4, 32, xyz, q, return_counts=True)
    ref_i, ref_c = ball_query(0.4, 32, xyz[-8:].contiguous(), q[-8:].contiguous(), return_counts=True)
    assert torch.equal(idx[-8:], ref_i) and torch.equal(cnt[-8:], ref_c)
    prims = {k: v[:4].repeat((Bq + 3) // 4, *([1] * (v.ndim - 1)))[:Bq].contiguous() for k, v in prob.items()
             if k.startswith(("cuboid_", "cylinder_"))}
    cloud = sample_scene_clouds(prims, 64, 5)
    tail = sample_scene_clouds({k: v[-4:].contiguous() for k, v in prims.items()}, 64, 5, env_offset=Bq - 4)
    assert torch.equal(cloud[-4:], tail)
