"""Pins the C oracle's geometry against golden vectors produced by the reference's own
mpinets/geometry.py (tests/golden/gen_geometry_golden.py).  CPU only."""
import numpy as np
import pytest

TOL = 2e-6  # oracle (fixed fp32 op order) vs torch CPU kernels; reference values are O(1)


def _suites(golden):
    return [str(s) for s in golden["suites"]]


def _g(golden, name, key):
    return golden[f"{name}/{key}"]


def test_suites_present(golden):
    assert len(_suites(golden)) == 6


@pytest.mark.parametrize("suite", ["tabletop_yaw", "cubby_yaw_padded", "full_rotation_quirk",
                                   "unnormalised_quats", "all_masked", "single_prim"])
def test_inverse_frames_match_reference(golden, oracle, suite):
    for kind in ("cub", "cyl"):
        got = oracle.inv_frames_4x4(_g(golden, suite, f"{kind}_centers"), _g(golden, suite, f"{kind}_quats"))
        ref = _g(golden, suite, f"out/{kind}_inv_frames")
        np.testing.assert_allclose(got, ref, rtol=0, atol=TOL)


def test_quirk_matrix_is_not_orthonormal(golden, oracle):
    """geometry.py:212-213 writes `yz - wx` twice; the oracle must reproduce it, not fix it."""
    f = oracle.inv_frames_4x4(_g(golden, "full_rotation_quirk", "cub_centers"),
                              _g(golden, "full_rotation_quirk", "cub_quats"))
    R = f[..., :3, :3]
    err = np.abs(R @ np.swapaxes(R, -1, -2) - np.eye(3)).max()
    assert err > 1e-2
    np.testing.assert_array_equal(R[..., 1, 2], R[..., 2, 1])


@pytest.mark.parametrize("suite", ["tabletop_yaw", "cubby_yaw_padded", "full_rotation_quirk",
                                   "unnormalised_quats", "all_masked", "single_prim"])
def test_sdf_matches_reference(golden, oracle, suite):
    g = lambda k: _g(golden, suite, k)
    for pts, tag in ((g("points"), "sdf"), (g("seq"), "sdf_seq")):
        got = oracle.cuboid_sdf(g("cub_centers"), g("cub_dims"), g("cub_quats"), pts)
        np.testing.assert_allclose(got, g(f"out/cub_{tag}"), rtol=0, atol=TOL)
        got = oracle.cylinder_sdf(g("cyl_centers"), g("cyl_radii"), g("cyl_heights"), g("cyl_quats"), pts)
        np.testing.assert_allclose(got, g(f"out/cyl_{tag}"), rtol=0, atol=TOL)
        got = oracle.sphere_sdf(g("sph_centers"), g("sph_radii"), pts)
        np.testing.assert_allclose(got, g(f"out/sph_{tag}"), rtol=0, atol=TOL)


def test_all_masked_is_inf(golden, oracle):
    g = lambda k: _g(golden, "all_masked", k)
    assert np.isinf(g("out/cub_sdf")).all() and np.isinf(g("out/cyl_sdf_seq")).all()
    assert np.isinf(oracle.cuboid_sdf(g("cub_centers"), g("cub_dims"), g("cub_quats"), g("points"))).all()
    assert np.isinf(oracle.sphere_sdf(g("sph_centers"), g("sph_radii"), g("seq"))).all()


def test_sequence_equals_per_slice(golden, oracle):
    """Property the reference satisfies: sdf_sequence(x)[:,t] == sdf(x[:,t])."""
    g = lambda k: _g(golden, "tabletop_yaw", k)
    seq = g("seq")
    whole = oracle.cylinder_sdf(g("cyl_centers"), g("cyl_radii"), g("cyl_heights"), g("cyl_quats"), seq)
    for t in range(seq.shape[1]):
        part = oracle.cylinder_sdf(g("cyl_centers"), g("cyl_radii"), g("cyl_heights"), g("cyl_quats"), seq[:, t])
        np.testing.assert_array_equal(whole[:, t], part)
    np.testing.assert_allclose(g("out/cyl_sdf_seq")[:, 0],
                               oracle.cylinder_sdf(g("cyl_centers"), g("cyl_radii"), g("cyl_heights"),
                                                   g("cyl_quats"), seq[:, 0]), atol=TOL)


def test_collision_flags_follow_model_py(golden, oracle):
    """model.py:293-314 restated with the oracle SDFs == direct evaluation from reference SDFs."""
    g = lambda k: _g(golden, "cubby_yaw_padded", k)
    seq = g("seq")  # [B,T,N,3] used as sphere centres
    S = seq.shape[2]
    radii = np.linspace(0.01, 0.4, S).astype(np.float32)
    flags, msdf = oracle.collision_flags(seq, radii, (g("cub_centers"), g("cub_dims"), g("cub_quats")),
                                         (g("cyl_centers"), g("cyl_radii"), g("cyl_heights"), g("cyl_quats")))
    ref_min = np.minimum(g("out/cub_sdf_seq"), g("out/cyl_sdf_seq"))
    np.testing.assert_allclose(msdf, ref_min, atol=TOL)
    ref_flags = (ref_min <= radii[None, None, :]).reshape(seq.shape[0], -1).any(axis=1)
    margin = np.abs(ref_min - radii[None, None, :]).min()
    assert margin > 10 * TOL  # the fixture has no borderline sphere
    np.testing.assert_array_equal(flags, ref_flags)


def test_joint_normalisation_arithmetic(golden, oracle):
    """mpinets/utils.py:91-93 and :207-209 (limits table is this repo's, arithmetic is the reference's)."""
    from mpinets_amd import franka_tables as ft

    got = oracle.unnormalize(golden["utils/q_norm"], ft.JOINT_LIMITS_REAL)
    np.testing.assert_allclose(got, golden["utils/unnormalized"], rtol=0, atol=1e-6)
    back = oracle.normalize(golden["utils/unnormalized"], ft.JOINT_LIMITS_REAL)
    np.testing.assert_allclose(back, golden["utils/renormalized"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(back, golden["utils/q_norm"], rtol=0, atol=1e-5)


def test_depth_render_known_answers(oracle):
    """Row N4 oracle: a camera at the origin looking along -z (identity pose) at a box / cylinder / robot sphere."""
    I = np.eye(4, dtype=np.float32)[None]
    W, H, f = 65, 49, 40.0
    intr = (f, f, W / 2.0, H / 2.0)
    quat = np.array([[[1.0, 0, 0, 0]]], np.float32)
    cub = (np.array([[[0, 0, -3.0]]], np.float32), np.array([[[1.0, 1.0, 2.0]]], np.float32), quat)
    none_cyl = (np.zeros((1, 1, 3), np.float32), np.zeros((1, 1, 1), np.float32), np.zeros((1, 1, 1), np.float32), quat)
    d = oracle.depth_render(I, intr, W, H, cub, none_cyl).reshape(H, W)
    c = d[H // 2, W // 2]  # centre pixel: straight down the axis, front face at z = -2
    assert abs(c - 2.0) < 1e-6
    assert d[0, 0] == -1 and (d >= 0).sum() > 100  # corners miss, the box fills the middle
    # a pixel off-axis hits the same front face: its ray length is 2 / cos(angle)
    u, v = W // 2 + 6, H // 2 - 4
    x, y = (u + 0.5 - intr[2]) / f, -(v + 0.5 - intr[3]) / f
    assert abs(d[v, u] - 2.0 * np.sqrt(1 + x * x + y * y)) < 1e-5
    # cylinder on the axis (its own z along the view direction): the near cap at distance 4 - 0.5
    cyl = (np.array([[[0, 0, -4.0]]], np.float32), np.array([[[0.3]]], np.float32), np.array([[[1.0]]], np.float32), quat)
    none_cub = (np.zeros((1, 1, 3), np.float32), np.zeros((1, 1, 3), np.float32), quat)
    assert abs(oracle.depth_render(I, intr, W, H, none_cub, cyl).reshape(H, W)[H // 2, W // 2] - 3.5) < 1e-6
    # a robot sphere in front of the box removes those pixels; behind the box it does not
    front = oracle.depth_render(I, intr, W, H, cub, none_cyl, np.array([[[0, 0, -1.0]]], np.float32), np.array([0.1], np.float32))
    back = oracle.depth_render(I, intr, W, H, cub, none_cyl, np.array([[[0, 0, -6.0]]], np.float32), np.array([0.1], np.float32))
    assert front.reshape(H, W)[H // 2, W // 2] == -1 and back.reshape(H, W)[H // 2, W // 2] == c
    pts, cnt = oracle.depth_select(d.reshape(1, -1), I, intr, W, H, 50, 3)
    assert cnt[0] == (d >= 0).sum() and np.allclose(pts[0, :, 2].max(), -2.0, atol=1e-5) and len(np.unique(pts[0], axis=0)) == 50
