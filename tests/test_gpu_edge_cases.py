"""Edge cases of the C-ABI entry points: ragged / tiny / empty sizes, unaligned views, tails."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


@pytest.mark.parametrize("B,N,npoint", [(1, 1, 1), (2, 5, 5), (3, 64, 1), (1, 65, 64), (2, 513, 100), (1, 4097, 31)])
def test_fps_small_and_ragged(oracle, B, N, npoint):
    from mpinets_amd.pointnet2 import furthest_point_sample

    x = np.random.default_rng(N).uniform(-1, 1, (B, N, 3)).astype(np.float32)
    idx = furthest_point_sample(T(x), npoint)
    np.testing.assert_array_equal(idx.cpu().numpy(), oracle.fps(x, npoint))


def test_fps_rejects_oversized_cloud():
    from mpinets_amd import _lib
    from mpinets_amd.pointnet2 import furthest_point_sample

    with pytest.raises(_lib.MpxError, match="8192"):
        furthest_point_sample(torch.zeros(1, 9000, 3, device=dev()), 4)


@pytest.mark.parametrize("N,npoint,nsample,stride", [(6271, 300, 64, 4), (1001, 257, 32, 4), (50, 3, 128, 3), (7, 7, 32, 4)])
def test_ball_query_unaligned_and_tails(oracle, N, npoint, nsample, stride):
    """N % 4 != 0 forces the generic (non 64-byte) slab path; npoint not a multiple of 256."""
    from mpinets_amd.pointnet2 import ball_query

    rng = np.random.default_rng(N + nsample)
    x = (rng.uniform(-1, 1, (2, N, stride)) * 0.4).astype(np.float32)
    c = np.ascontiguousarray(x[:, rng.permutation(N)[:npoint], :3])
    idx, cnt = ball_query(0.25, nsample, T(x), T(c), return_counts=True)
    ref, rcnt = oracle.ball_query(c, x, 0.25, nsample, return_counts=True)
    np.testing.assert_array_equal(idx.cpu().numpy(), ref)
    np.testing.assert_array_equal(cnt.cpu().numpy(), rcnt)


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("npoint,nsample,radius", [(37, 64, 0.1), (5, 32, 0.05), (130, 128, 2.0)])
def test_sa_module_tails_full_and_sparse_neighbourhoods(oracle, precision, npoint, nsample, radius):
    """Query counts that are not multiples of the per-wave group, nsample 32/64/128, neighbourhoods that are
    completely full (radius 2.0) or nearly empty (radius 0.05)."""
    from mpinets_amd.pointnet2 import PointnetSAModule

    torch.manual_seed(3)
    mod = PointnetSAModule(npoint=npoint, radius=radius, nsample=nsample, mlp=[1, 64, 64, 64], bn=False,
                           precision=precision).to(dev())
    rng = np.random.default_rng(npoint)
    xyz = (rng.uniform(-1, 1, (3, 700, 3)) * 0.5).astype(np.float32)
    feat = rng.normal(size=(3, 1, 700)).astype(np.float32)
    with torch.no_grad():
        nx, nf = mod(T(xyz), T(feat))
        mod.elide_padding = False
        _, nf_all = mod(T(xyz), T(feat))
    assert torch.equal(nf, nf_all)  # distinct-neighbour evaluation never changes a bit
    layers = [(c.weight.detach().cpu().numpy(), c.bias.detach().cpu().numpy()) for c in mod.convs()]
    onx, onf, _ = oracle.sa_module(xyz, feat, npoint, radius, nsample, layers)
    np.testing.assert_array_equal(nx.cpu().numpy(), onx)
    tol = 1e-5 if precision == "fp32" else 1e-4
    np.testing.assert_allclose(nf.cpu().numpy(), onf, rtol=tol, atol=tol)


def test_sa_with_queries_that_hit_nothing(oracle):
    """A query with no point in range gets the zero-initialised index row: 128 copies of point 0."""
    from mpinets_amd import _lib
    from mpinets_amd.pointnet2 import PointnetSAModule, ball_query, sa_mlp_fused

    torch.manual_seed(1)
    mod = PointnetSAModule(npoint=8, radius=0.1, nsample=32, mlp=[1, 64, 64, 64], bn=False).to(dev())
    rng = np.random.default_rng(0)
    xyz = T((rng.uniform(-1, 1, (1, 200, 4)) * 0.3).astype(np.float32))
    centres = xyz[:, :8, :3].clone().contiguous()
    centres[:, 5] = 40.0  # far away
    nbr, cnt = ball_query(0.1, 32, xyz, centres, return_counts=True)
    assert int(cnt[0, 5]) == 0 and (nbr[0, 5] == 0).all()
    wpack = mod._packed.get(mod.convs(), 1, "fp32")
    a = sa_mlp_fused(xyz, centres, xyz[:, :, 3:], 4, 1, nbr, wpack, (64, 64, 64), cnt=cnt)
    b = sa_mlp_fused(xyz, centres, xyz[:, :, 3:], 4, 1, nbr, wpack, (64, 64, 64), cnt=None)
    assert torch.equal(a, b)


def test_sdf_without_primitives_and_empty_batches():
    from mpinets_amd.geometry import TorchCuboids, TorchCylinders
    from mpinets_amd.robot import FrankaCollisionSampler

    pts = torch.rand(2, 9, 3, device=dev())
    cub = TorchCuboids(torch.zeros(2, 0, 3, device=dev()), torch.zeros(2, 0, 3, device=dev()), torch.zeros(2, 0, 4, device=dev()))
    assert torch.isinf(cub.sdf(pts)).all()
    cyl = TorchCylinders(torch.zeros(2, 0, 3, device=dev()), torch.zeros(2, 0, 1, device=dev()), torch.zeros(2, 0, 1, device=dev()),
                         torch.zeros(2, 0, 4, device=dev()))
    assert torch.isinf(cyl.sdf_sequence(pts[:, None])).all()
    assert cub.sdf(torch.zeros(2, 0, 3, device=dev())).shape == (2, 0)
    cs = FrankaCollisionSampler(dev())
    q = torch.zeros(0, 3, 7, device=dev())
    assert cs.check(q, None, None).shape == (0,)
    assert not cs.check(torch.zeros(4, 2, 7, device=dev()), cub.__class__(torch.zeros(4, 0, 3, device=dev()), torch.zeros(4, 0, 3, device=dev()),
                                                                       torch.zeros(4, 0, 4, device=dev())), None).any()


def test_linear_tiny_and_unaligned_outputs():
    from mpinets_amd.pointnet2 import linear

    rng = np.random.default_rng(5)
    for M, N, K in [(1, 1, 4), (3, 7, 8), (129, 130, 20), (5, 3, 2112)]:
        x, w, b = (T(rng.normal(size=s).astype(np.float32)) for s in ((M, K), (N, K), (N,)))
        ref = x.double() @ w.double().T + b.double()
        y = linear(x, w, b, 0)
        assert (y.double() - ref).abs().max() < 1e-4 * np.sqrt(K)
        wide = torch.full((M, N + 3), 9.0, device=dev())  # odd leading dimension + column offset: scalar-store path
        linear(x, w, b, 0, out=wide[:, 1:1 + N])
        assert torch.equal(wide[:, 1:1 + N], y) and (wide[:, 0] == 9).all() and (wide[:, 1 + N:] == 9).all()


def test_scene_cloud_odd_sizes(oracle):
    from mpinets_amd.scenes import make_scenes, sample_scene_clouds

    scn = make_scenes(3, 2, ("tabletop",), 5, 2)
    prims = {k: T(v) for k, v in scn.items()}
    out, assign, labels, nobs = sample_scene_clouds(prims, 1001, seed=99, return_aux=True)
    opts, oassign, olabels, onobs = oracle.scene_cloud(scn, 1001, 99)
    np.testing.assert_array_equal(assign.cpu().numpy().view(np.uint16), oassign)
    np.testing.assert_allclose(out.cpu().numpy(), opts, atol=1e-6)


def test_stream_without_a_unit_queue_slot_never_shares_one():
    """MPX_VARIANT_UNIT_QUEUE = 0 stands in for "256 distinct streams already seen": a NEW stream gets no counters.  The
    fp32 grouped MLPs then run one unit per wave without a queue -- bit-identical output -- and the bf16x3 persistent
    kernel, which cannot run without one, reports it instead of sharing another stream's counters (round-3 advisor)."""
    from mpinets_amd import _lib
    from mpinets_amd.model import MotionPolicyNetwork
    from mpinets_amd.scenes import make_problem_batch

    lib = _lib.load()
    torch.manual_seed(3)
    mdl = MotionPolicyNetwork().to(dev()).eval()
    prob = make_problem_batch(640, seed=4, device=dev(), scene_pool=8)  # (>= 4 units per wave slot: the persistent launches)
    with torch.no_grad():
        ref = mdl(prob["xyz"], prob["q_norm"])
        torch.cuda.synchronize()
        assert lib.mpx_set_variant(2, 0) == 0 and lib.mpx_get_variant(2) == 0
        try:
            fresh = torch.cuda.Stream(device=dev())
            with torch.cuda.stream(fresh):
                got = mdl(prob["xyz"], prob["q_norm"])
                mdl.set_precision("bf16x3")
                with pytest.raises(_lib.MpxError, match="no unit-queue slot"):
                    mdl(prob["xyz"], prob["q_norm"])
            fresh.synchronize()
        finally:
            mdl.set_precision("fp32")
            assert lib.mpx_set_variant(2, 1) == 0
    assert torch.equal(got, ref)


def test_ball_query_hits_rejects_rows_longer_than_the_counts_are_honoured():
    from mpinets_amd import _lib

    x = torch.zeros(1, 600, 3, device=dev())
    idx = torch.empty((1, 4, 288), dtype=torch.int32, device=dev())
    cnt = torch.empty((1, 4), dtype=torch.int32, device=dev())
    with pytest.raises(_lib.MpxError, match="nsample = 288 > 256"):
        _lib.call("mpx_ball_query_hits", _lib.ptr(x), 3, _lib.ptr(x), 3, 1, 600, 4, 0.1, 288, _lib.ptr(idx), _lib.ptr(cnt))


def test_policy_forward_without_the_group_all_pack():
    """struct mpx_policy_weights.sa3_pack = NULL: the layer-by-layer group-all module at every batch size."""
    import ctypes

    from mpinets_amd import _lib
    from mpinets_amd.model import MotionPolicyNetwork
    from mpinets_amd.scenes import make_problem_batch

    torch.manual_seed(5)
    mdl = MotionPolicyNetwork().to(dev()).eval()
    prob = make_problem_batch(256, seed=6, device=dev(), scene_pool=4)
    xyz, q = prob["xyz"], prob["q_norm"]
    with torch.no_grad():
        ref = mdl.forward_native(xyz, q)
        w, keep = mdl.native_weights()
        w.sa3_pack = None
        need = _lib.load().mpx_policy_workspace(256, xyz.size(1))
        ws = torch.empty(need, dtype=torch.uint8, device=dev())
        dq = torch.empty((256, 7), dtype=torch.float32, device=dev())
        _lib.call("mpx_policy_forward", ctypes.addressof(w), _lib.ptr(xyz), xyz.size(1), _lib.ptr(q), 256, _lib.ptr(dq),
                  _lib.ptr(ws), need)
    assert (dq - ref).abs().max().item() <= 1e-6
