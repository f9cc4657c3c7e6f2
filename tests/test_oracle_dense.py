"""CPU: the oracle vs the reference's model.py run on DENSE clouds (tests/golden/gen_dense_golden.py).

Two of the three environments overflow both ball queries (hundreds of hits for 128 slots): the truncation rule -- the
first ``nsample`` hits by index -- and full neighbourhood rows go through the reference's own grouping, shared MLPs and
max-pools here, not only through sparse rows as in model_golden.npz."""
import numpy as np

from test_oracle_model import NR, golden_state_dict

TOL = 1e-6


def test_the_dense_environments_overflow_both_modules(dense_golden):
    g = dense_golden
    h1, h2 = g["d_hits1"].astype(int), g["d_hits2"].astype(int)
    assert ((h1 > 128).sum(1)[:2] >= 20).all() and ((h2 > 128).sum(1)[:2] >= 100).all()
    assert (h1[2] <= 128).all() and (h2[2] <= 128).all()  # the sparse row of the batch
    # an overflowing row holds 128 DISTINCT ascending indices (no padding), a sparse one repeats its first hit
    b1 = g["d_ball1"].astype(int)
    full = h1 > 128
    assert (np.diff(b1[full], axis=1) > 0).all()
    sparse = b1[~full & (h1 > 0) & (h1 < 128)]
    assert (sparse[:, -1] == sparse[:, 0]).all()


def test_indices_under_overflow(oracle, dense_golden):
    """The index path on the golden's own clouds: FPS picks, and ball-query rows equal to the brute-force first-128 rule."""
    g = dense_golden
    xyz = np.ascontiguousarray(g["d_xyz"][:, :, :3])
    np.testing.assert_array_equal(oracle.fps(xyz, 512), g["d_fps1"])
    np.testing.assert_array_equal(oracle.fps(g["d_xyz1"], 128), g["d_fps2"])
    for new, pts, r, key in ((g["d_xyz1"], xyz, 0.05, "d_ball1"), (g["d_xyz2"], g["d_xyz1"], 0.3, "d_ball2")):
        got = oracle.ball_query(new, pts, r, 128)
        np.testing.assert_array_equal(got, g[key].astype(np.int32))
        # brute force, in float64 away from the radius: the first 128 indices inside the ball, padded with the first
        d2 = ((new[:, :, None, :].astype(np.float64) - pts[:, None, :, :].astype(np.float64)) ** 2).sum(-1)
        for b, j in ((0, 0), (0, 5), (1, 17), (1, 100), (2, 3)):
            inside = np.flatnonzero(d2[b, j] < r * r - 1e-7)
            maybe = np.flatnonzero(d2[b, j] < r * r + 1e-7)
            if len(inside) != len(maybe):
                continue  # a point on the sphere to within rounding: float32 decides, not this check
            want = np.full(128, inside[0] if len(inside) else 0)
            want[:min(128, len(inside))] = inside[:128]
            np.testing.assert_array_equal(got[b, j], want)


def test_forward_and_every_module_output(oracle, dense_golden, model_golden):
    import seeded_weights

    g = dense_golden
    sd = golden_state_dict(model_golden)  # (the same seeded weights as model_golden.npz: the digest is in both files)
    assert seeded_weights.digest(sd) == str(g["param_sha256"])
    dq, aux = oracle.policy_forward(sd, g["d_xyz"], g["d_q"])
    np.testing.assert_array_equal(aux["sa1"]["fps_idx"], g["d_fps1"])
    np.testing.assert_array_equal(aux["sa2"]["fps_idx"], g["d_fps2"])
    np.testing.assert_array_equal(aux["xyz2"], g["d_xyz2"])
    for mine, ref in ((aux["f1"], g["d_feat1"]), (aux["f2"], g["d_feat2"]), (aux["f3"], g["d_feat3"]),
                      (aux["encoding"], g["d_encoding"]), (dq, g["d_out"])):
        assert mine.shape == ref.shape
        np.testing.assert_allclose(mine, ref, rtol=0, atol=2 * TOL)


def test_rollout_from_the_dense_slabs(oracle, dense_golden, model_golden):
    from mpinets_amd import franka_tables as ft

    g = dense_golden
    sd = golden_state_dict(model_golden)
    pts, link = ft.link_point_table(4096, True)
    slab = g["d_xyz"].copy()
    sampler = lambda qu, i: oracle.transform_table(oracle.franka_fk(qu), pts, link, g["d_subsets"][i])
    traj = oracle.rollout(sd, slab, g["d_q"], 5, sampler, ft.JOINT_LIMITS_REAL)
    np.testing.assert_allclose(np.stack(traj), g["d_traj"], rtol=0, atol=5 * TOL)
    np.testing.assert_allclose(slab[:, :NR, :3], g["d_robot"], rtol=0, atol=5 * TOL)
    np.testing.assert_array_equal(slab[:, NR:], g["d_xyz"][:, NR:])
