"""FPS / ball query / grouping: bit-exact index parity with the oracle (pointnet2_ops semantics)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


def clouds(B, N, seed, stride=3):
    rng = np.random.default_rng(seed)
    x = rng.uniform(-1, 1, (B, N, stride)).astype(np.float32)
    return x


@pytest.mark.parametrize("B,N,npoint,stride", [(3, 6272, 512, 4), (4, 512, 128, 3), (2, 1000, 64, 3),
                                               (2, 63, 17, 3), (1, 8192, 300, 4), (2, 2500, 100, 3),
                                               (2, 513, 40, 3), (2, 4097, 77, 3), (1, 7000, 7000, 3)])
def test_fps_bit_exact(oracle, B, N, npoint, stride):
    from mpinets_amd.pointnet2 import furthest_point_sample

    x = clouds(B, N, N + npoint, stride)
    idx, nx = furthest_point_sample(T(x), npoint, return_xyz=True)
    ref = oracle.fps(x, npoint)
    np.testing.assert_array_equal(idx.cpu().numpy(), ref)
    np.testing.assert_array_equal(nx.cpu().numpy(), oracle.gather_points(x, ref))


def test_fps_ties_duplicates_and_skipped_points(oracle):
    """Exact ties (lattice points, duplicated points) exercise the (k mod bs, k) tie order; points
    with |p|^2 <= 1e-3 must be skipped; an all-skipped cloud returns zeros."""
    from mpinets_amd.pointnet2 import furthest_point_sample

    rng = np.random.default_rng(5)
    g = np.stack(np.meshgrid(*[np.arange(-7, 8)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.float32) * 0.125
    lattice = g[rng.permutation(len(g))][:3000]
    dup = clouds(1, 1500, 1)[0]
    dup = np.concatenate([dup, dup[::-1]], 0)  # every point twice
    near0 = clouds(1, 3000, 2)[0]
    near0[::3] *= 0.01  # a third of the points inside the skipped ball
    x = np.stack([lattice, dup, near0])
    x[2, 0] = [0.001, 0.0, 0.0]  # the start index itself is a skipped point
    idx = furthest_point_sample(T(x), 256)
    np.testing.assert_array_equal(idx.cpu().numpy(), oracle.fps(x, 256))
    z = np.zeros((2, 700, 3), np.float32)
    z[1] = 0.01
    np.testing.assert_array_equal(furthest_point_sample(T(z), 9).cpu().numpy(), oracle.fps(z, 9))
    assert (oracle.fps(z, 9) == 0).all()


def test_fps_degenerate_clouds_through_the_culled_kernel(oracle):
    """Clouds whose bounding box has no extent along one, two or all axes (the Morton cells of the culled kernel
    collapse), clouds of a few distinct points repeated thousands of times, and one far outlier."""
    from mpinets_amd.pointnet2 import furthest_point_sample

    rng = np.random.default_rng(12)
    N = 3000
    plane = clouds(1, N, 3)[0]
    plane[:, 2] = 0.25
    line = np.zeros((N, 3), np.float32)
    line[:, 0] = rng.uniform(-1, 1, N)
    line[:, 1] = 0.5
    same = np.full((N, 3), 0.3, np.float32)
    few = clouds(1, 7, 4)[0][rng.integers(0, 7, N)]
    outlier = clouds(1, N, 5)[0] * 0.05 + 0.5
    outlier[1234] = [90.0, -80.0, 70.0]
    x = np.stack([plane, line, same, few, outlier])
    idx = furthest_point_sample(T(x), 200)
    np.testing.assert_array_equal(idx.cpu().numpy(), oracle.fps(x, 200))


def test_fps_on_real_slab(oracle):
    """Scene-like slab rows (robot | scene | target) at the model's sizes."""
    from mpinets_amd.pointnet2 import furthest_point_sample
    from mpinets_amd.scenes import make_problem_batch

    prob = make_problem_batch(3, seed=4, device=dev(), kinds=("tabletop", "cubby", "dresser"), M1=40)
    xyz = prob["xyz"]
    idx = furthest_point_sample(xyz, 512)
    np.testing.assert_array_equal(idx.cpu().numpy(), oracle.fps(xyz.cpu().numpy(), 512))


@pytest.mark.parametrize("B,N,npoint,radius,nsample,stride", [(2, 6272, 512, 0.05, 128, 4), (3, 512, 128, 0.3, 128, 3),
                                                              (2, 700, 50, 0.4, 32, 3), (1, 100, 10, 5.0, 64, 3),
                                                              (2, 300, 70, 0.5, 16, 4), (1, 512, 129, 0.9, 128, 3),
                                                              (3, 65, 65, 0.2, 128, 3)])
def test_ball_query_bit_exact(oracle, B, N, npoint, radius, nsample, stride):
    from mpinets_amd.pointnet2 import ball_query

    x = clouds(B, N, 17 + N, stride)
    x[..., :3] *= 0.5
    centres = np.ascontiguousarray(x[:, :npoint, :3]).copy()
    centres[:, -1] = 50.0  # a query with no neighbour at all -> zeros
    idx, cnt = ball_query(radius, nsample, T(x), T(centres), return_counts=True)
    ref, rcnt = oracle.ball_query(centres, x, radius, nsample, return_counts=True)
    np.testing.assert_array_equal(idx.cpu().numpy(), ref)
    np.testing.assert_array_equal(cnt.cpu().numpy(), rcnt)
    assert (ref[:, -1] == 0).all() and (rcnt[:, -1] == 0).all()


@pytest.mark.parametrize("N,npoint,stride", [(2048, 300, 3), (4096, 512, 4), (6272, 512, 4), (2047, 128, 4), (8192, 700, 3)])
def test_ball_query_bucketed_path_edge_cases(oracle, N, npoint, stride):
    """The large-cloud / small-radius path (column grid in LDS) must reproduce the reference order exactly:
    a dense cluster (> nsample hits: the nsample SMALLEST indices), points far outside the grid extent, duplicated
    points, points exactly on a cell border, empty neighbourhoods.  N = 2047 takes the brute-force kernel."""
    from mpinets_amd.pointnet2 import ball_query

    rng = np.random.default_rng(N + npoint)
    B, r, ns = 3, 0.05, 128
    x = np.zeros((B, N, stride), np.float32)
    x[..., :3] = rng.uniform(-0.6, 0.9, (B, N, 3))
    x[:, 100:500, :3] = np.float32([0.3, -0.2, 0.4]) + rng.normal(scale=0.012, size=(B, 400, 3))  # dense: ~400 hits
    x[:, 500:520, :3] = x[:, 100:120, :3]                                    # duplicates of cluster points
    x[:, 600:620, :3] = rng.uniform(5.0, 9.0, (B, 20, 3))                    # far outside the 48-column extent
    x[:, 620:640, 0] = x[:, :1, 0].min() + 0.05 * 1.0001 * np.arange(20)     # on column borders
    x[:, N - 5:, :3] = np.float32([-3.0, -3.0, 0.0])                         # moves the grid origin far away
    centres = np.ascontiguousarray(x[:, rng.permutation(N)[:npoint], :3]).copy()
    centres[:, 0] = [0.3, -0.2, 0.4]          # centre of the dense cluster
    centres[:, 1] = x[:, 605, :3]             # in the far-away group
    centres[:, 2] = 50.0                      # nothing around
    centres[:, 3] = x[:, 625, :3]
    idx, cnt = ball_query(r, ns, T(x), T(centres), return_counts=True)
    ref, rcnt = oracle.ball_query(centres, x, r, ns, return_counts=True)
    assert rcnt[:, 0].min() == ns and rcnt[:, 2].max() == 0
    np.testing.assert_array_equal(cnt.cpu().numpy(), rcnt)
    np.testing.assert_array_equal(idx.cpu().numpy(), ref)


def test_ball_query_bucketed_path_smaller_nsample(oracle):
    """nsample 64 on the bucketed path (rows never exceed one key per lane)."""
    from mpinets_amd.pointnet2 import ball_query

    rng = np.random.default_rng(5)
    x = rng.uniform(-0.4, 0.4, (2, 3000, 3)).astype(np.float32)
    x[:, :200] = rng.normal(scale=0.02, size=(2, 200, 3))  # > 64 hits around the origin
    centres = np.ascontiguousarray(x[:, ::10][:, :256]).copy()
    centres[:, 0] = 0.0
    idx, cnt = ball_query(0.06, 64, T(x), T(centres), return_counts=True)
    ref, rcnt = oracle.ball_query(centres, x, 0.06, 64, return_counts=True)
    assert rcnt[:, 0].min() == 64
    np.testing.assert_array_equal(cnt.cpu().numpy(), rcnt)
    np.testing.assert_array_equal(idx.cpu().numpy(), ref)


def test_group_points_matches_oracle(oracle):
    from mpinets_amd.pointnet2 import ball_query, furthest_point_sample, query_and_group

    x = clouds(2, 900, 3)
    feat = np.random.default_rng(1).normal(size=(2, 5, 900)).astype(np.float32)
    idx, nx = furthest_point_sample(T(x), 40, return_xyz=True)
    nbr = ball_query(0.4, 32, T(x), nx)
    got = query_and_group(T(x), nx, T(np.ascontiguousarray(feat.transpose(0, 2, 1))), nbr)
    ref = oracle.group_points(x, nx.cpu().numpy(), feat, nbr.cpu().numpy())
    np.testing.assert_array_equal(got.cpu().numpy(), ref)


def test_sort_queries_by_tiles():
    from mpinets_amd import _lib

    rng = np.random.default_rng(0)
    cnt = rng.integers(0, 140, 5000).astype(np.int32)
    tc = T(cnt)
    order = torch.empty(5000, dtype=torch.int32, device=dev())
    scratch = torch.empty(128, dtype=torch.int32, device=dev())
    _lib.call("mpx_sort_queries", _lib.ptr(tc), 5000, 128, _lib.ptr(order), _lib.ptr(scratch))
    o = order.cpu().numpy()
    assert sorted(o.tolist()) == list(range(5000))  # a permutation
    rows = (np.clip(cnt, 1, 128) + 3) // 4 * 4
    assert (np.diff(rows[o]) <= 0).all()  # non-increasing packed row count


def _threshold_cloud():
    """Points whose squared norm is exactly 1e-3f (kept upstream: the float is compared with the DOUBLE literal
    1e-3 < 1e-3f) and one float below it (skipped)."""
    from test_oracle_pointnet import threshold_points

    return threshold_points()


def test_fps_skip_threshold_is_the_double_comparison(oracle):
    from mpinets_amd.pointnet2 import furthest_point_sample

    x, expect = _threshold_cloud()
    ref = oracle.fps(x, expect.shape[1])
    np.testing.assert_array_equal(ref, expect)
    np.testing.assert_array_equal(furthest_point_sample(T(x), expect.shape[1]).cpu().numpy(), ref)


@pytest.mark.parametrize("N,npoint", [(1, 1), (2, 2), (63, 40), (64, 64), (65, 33), (127, 100), (128, 128), (300, 77), (511, 128),
                                      (512, 128), (512, 512)])
def test_fps_small_clouds_one_wave_per_environment(oracle, N, npoint):
    """N <= 512 takes fps_wave_kernel (one wave per environment, cloud in registers): every lane occupancy from a single
    point to all 8 register slots, with duplicated points (exact ties), lattice points and skipped points mixed in."""
    from mpinets_amd.pointnet2 import furthest_point_sample

    rng = np.random.default_rng(N * 1000 + npoint)
    x = rng.uniform(-1, 1, (5, N, 3)).astype(np.float32)
    x[1] = np.round(x[1] * 4) / 4                      # lattice: many exact ties
    if N >= 4:
        x[2, N // 2:] = x[2, :N - N // 2]              # duplicates
        x[3, ::3] *= 0.01                              # a third of the points inside the skipped ball
    x[4] = 0.001                                       # nothing is a candidate: all indices 0
    idx, nx = furthest_point_sample(T(x), npoint, return_xyz=True)
    ref = oracle.fps(x, npoint)
    np.testing.assert_array_equal(idx.cpu().numpy(), ref)
    np.testing.assert_array_equal(nx.cpu().numpy(), oracle.gather_points(x, ref))


@pytest.mark.parametrize("N,npoint,nsample", [(2048, 513, 64), (3000, 1000, 128), (8192, 37, 128), (6272, 4096, 96), (5000, 7, 128)])
def test_ball_query_bucketed_ragged_query_counts(oracle, N, npoint, nsample):
    """Query counts that do not divide the workgroup's 16 waves / two lanes per query evenly, incl. a dense cluster that
    overflows nsample (redone in index order)."""
    from mpinets_amd.pointnet2 import ball_query

    rng = np.random.default_rng(N + npoint)
    xyz = rng.uniform(-0.6, 0.6, (2, N, 3)).astype(np.float32)
    xyz[1, : N // 8] = xyz[1, 0] + rng.normal(0, 0.01, (N // 8, 3)).astype(np.float32)  # > nsample hits around one spot
    q = xyz[:, rng.permutation(N)[:npoint] if npoint <= N else rng.integers(0, N, npoint)].copy()
    q[:, 0] = xyz[:, 0]  # (environment 1: the centre of the dense cluster)
    idx, cnt = ball_query(0.05, nsample, T(xyz), T(q), return_counts=True)
    ridx, rcnt = oracle.ball_query(q, xyz, 0.05, nsample, return_counts=True)
    np.testing.assert_array_equal(cnt.cpu().numpy(), rcnt)
    np.testing.assert_array_equal(idx.cpu().numpy(), ridx)
    assert rcnt.max() == nsample and rcnt.min() >= 1


def test_contraction_order_ab_on_the_device(oracle, tmp_path):
    """Both sides of the contraction-order A/B are real: the library built with -DMPX_SQDIST_XFIRST (the order rounds 1-2
    assumed; csrc/Makefile target `xfirst`, loaded in a subprocess through MPX_LIB_PATH) equals the oracle in ITS order 1
    bit for bit, like the product library equals the oracle in order 0 -- and the two libraries disagree on exactly the
    clouds on which the two oracle orders disagree."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    alt = os.path.join(root, "motion-policy-networks_amd", "mpinets_amd", "libmpinets_hip_xfirst.so")
    if not os.path.exists(alt):
        pytest.skip("libmpinets_hip_xfirst.so not built (make -C csrc xfirst)")
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_oracle_pointnet import _bench_like_clouds

    n = 256
    x = _bench_like_clouds(n, 23)
    np.save(tmp_path / "x.npy", x)
    code = ("import sys, numpy as np, torch; sys.path[:0] = [%r, %r]; from mpinets_amd.pointnet2 import furthest_point_sample, "
            "ball_query; x = torch.from_numpy(np.load(%r)).cuda(); i, c = furthest_point_sample(x, 512, return_xyz=True); "
            "b = ball_query(0.05, 128, x, c[:, :128].contiguous()); np.save(%r, i.cpu().numpy()); np.save(%r, b.cpu().numpy())")
    out = {}
    for tag, lib in (("product", None), ("xfirst", alt)):
        env = dict(os.environ)
        env.pop("MPX_LIB_PATH", None)
        if lib:
            env["MPX_LIB_PATH"] = lib
        fi, fb = str(tmp_path / f"{tag}_i.npy"), str(tmp_path / f"{tag}_b.npy")
        r = subprocess.run([sys.executable, "-c", code % (root, os.path.join(root, "motion-policy-networks_amd"),
                                                           str(tmp_path / "x.npy"), fi, fb)], env=env, capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        out[tag] = (np.load(fi), np.load(fb))
    for tag, order in (("product", 0), ("xfirst", 1)):
        oracle.set_sqdist_order(order)
        try:
            ref_i = oracle.fps(x, 512)
            ref_b = oracle.ball_query(oracle.gather_points(x, ref_i)[:, :128], x, 0.05, 128)
        finally:
            oracle.set_sqdist_order(0)
        np.testing.assert_array_equal(out[tag][0], ref_i, err_msg=tag)
        np.testing.assert_array_equal(out[tag][1], ref_b, err_msg=tag)


@pytest.mark.parametrize("N,npoint,radius,nsample,stride", [(6272, 512, 0.05, 128, 4), (512, 128, 0.3, 128, 3), (2047, 128, 0.05, 128, 4),
                                                            (3000, 1000, 0.08, 64, 3), (300, 70, 0.5, 16, 4)])
def test_ball_query_hits_only_entry_point(oracle, N, npoint, radius, nsample, stride):
    """mpx_ball_query_hits writes the hit slots of every row and nothing else: slots [0, max(cnt, 1)) and the counts equal
    the oracle's (an empty row holds index 0 in slot 0), the remaining slots keep the sentinel the buffer was filled
    with.  The fused grouped-MLP kernels read exactly those slots."""
    from mpinets_amd import _lib

    rng = np.random.default_rng(N + nsample)
    B = 3
    x = np.zeros((B, N, stride), np.float32)
    x[..., :3] = rng.uniform(-0.5, 0.5, (B, N, 3))
    x[:, 50:50 + 2 * nsample, :3] = np.float32([0.1, 0.2, -0.1]) + rng.normal(scale=radius * 0.2, size=(B, 2 * nsample, 3))
    centres = np.ascontiguousarray(x[:, rng.permutation(N)[:npoint], :3]).copy()
    centres[:, 0] = [0.1, 0.2, -0.1]   # more than nsample hits
    centres[:, 1] = 50.0               # none
    xd, cd = T(x), T(centres)
    idx = torch.full((B, npoint, nsample), -7, dtype=torch.int32, device=dev())
    cnt = torch.full((B, npoint), -7, dtype=torch.int32, device=dev())
    _lib.call("mpx_ball_query_hits", _lib.ptr(cd), 3, _lib.ptr(xd), stride, B, N, npoint, float(radius), nsample, _lib.ptr(idx),
              _lib.ptr(cnt), _lib.stream_ptr())
    ref, rcnt = oracle.ball_query(centres, x, radius, nsample, return_counts=True)
    got, gcnt = idx.cpu().numpy(), cnt.cpu().numpy()
    np.testing.assert_array_equal(gcnt, rcnt)
    assert rcnt[:, 0].min() == nsample and rcnt[:, 1].max() == 0
    live = np.arange(nsample)[None, None, :] < np.maximum(rcnt, 1)[..., None]
    np.testing.assert_array_equal(got[live], ref[live])
    assert (got[~live] == -7).all()
