"""Grouped MLP, dense layers, full policy forward and rollout vs the numpy oracle (fp32, 1e-5)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 1e-5  # BASELINE.json: "fp32 SDF and policy deltas within 1e-5"


def dev():
    return torch.device("cuda:0")


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


def test_mfma_operand_layout_probe():
    """mpx_linear on asymmetric data: catches a transposed / mis-mapped MFMA fragment."""
    from mpinets_amd.pointnet2 import linear

    rng = np.random.default_rng(0)
    for (M, N, K) in [(32, 32, 16), (128, 128, 16), (130, 7, 8), (257, 200, 260), (5, 4096, 1024), (1000, 64, 2112)]:
        x = rng.normal(size=(M, K)).astype(np.float32)
        w = rng.normal(size=(N, K)).astype(np.float32)
        b = rng.normal(size=N).astype(np.float32)
        for act in (0, 1, 2):
            y = linear(T(x), T(w), T(b), act).cpu().numpy()
            ref = x.astype(np.float64) @ w.astype(np.float64).T + b
            if act == 1:
                ref = np.maximum(ref, 0)
            if act == 2:
                ref = np.where(ref >= 0, ref, 0.01 * ref)
            scale = np.sqrt(K)
            assert np.abs(y - ref).max() <= 1e-5 * scale, (M, N, K, act, np.abs(y - ref).max())


def test_linear_strided_output_and_input():
    from mpinets_amd.pointnet2 import linear

    rng = np.random.default_rng(1)
    xw = T(rng.normal(size=(70, 40)).astype(np.float32))
    w = T(rng.normal(size=(24, 32)).astype(np.float32))
    out = torch.full((70, 100), -5.0, device=dev())
    linear(xw[:, :32], w, None, 0, out=out[:, 8:32])
    ref = xw[:, :32].double() @ w.double().T
    assert (out[:, 8:32].double() - ref).abs().max() < 1e-4
    assert (out[:, :8] == -5).all() and (out[:, 32:] == -5).all()


def test_linear_split_k_for_skinny_problems():
    """mpx_linear_ws: the K range of a layer with few output tiles is cut over the CUs; slices are added in
    order (deterministic), so the result matches float64 like the unsplit kernel and repeats bit for bit."""
    from mpinets_amd import _lib
    from mpinets_amd.pointnet2 import linear

    lib = _lib.load()
    assert lib.mpx_linear_workspace(8192, 4096, 1024) == 0  # enough tiles already
    assert lib.mpx_linear_workspace(9, 64, 128) == 0        # K too short to be worth a second launch
    assert lib.mpx_linear_workspace(1, 2048, 4096) == 0     # a few rows: the weight-streaming kernel, no split
    rng = np.random.default_rng(5)
    for (M, N, K) in [(9, 2048, 4096), (12, 512, 2112), (256, 2048, 2048), (50, 7, 256), (130, 300, 1000)]:
        need = lib.mpx_linear_workspace(M, N, K)
        assert need > 0 and need % (4 * M * N) == 0, (M, N, K, need)
        xw = T(rng.normal(size=(M, K + 12)).astype(np.float32))
        w = T(rng.normal(size=(N, K)).astype(np.float32))
        b = T(rng.normal(size=N).astype(np.float32))
        out = torch.full((M, N + 9), -5.0, device=dev())
        for act in (0, 2):
            linear(xw[:, :K], w, b, act, out=out[:, 4:4 + N])
            ref = xw[:, :K].double() @ w.double().T + b.double()
            if act == 2:
                ref = torch.where(ref >= 0, ref, 0.01 * ref)
            assert (out[:, 4:4 + N].double() - ref).abs().max() <= 1e-5 * np.sqrt(K), (M, N, K, act)
            assert (out[:, :4] == -5).all() and (out[:, 4 + N:] == -5).all()
            again = torch.empty((M, N), device=dev())
            linear(xw[:, :K], w, b, act, out=again)
            assert torch.equal(again, out[:, 4:4 + N])
        # a NULL workspace is the unsplit kernel
        y0 = torch.empty((M, N), device=dev())
        _lib.call("mpx_linear_ws", _lib.ptr(xw), xw.stride(0), _lib.ptr(w), _lib.ptr(b), M, N, K, 0, _lib.ptr(y0), N, None, 0)
        y1 = torch.empty((M, N), device=dev())
        _lib.call("mpx_linear", _lib.ptr(xw), xw.stride(0), _lib.ptr(w), _lib.ptr(b), M, N, K, 0, _lib.ptr(y1), N)
        assert torch.equal(y0, y1)
        # too small a workspace is refused
        small = torch.empty(16, dtype=torch.uint8, device=dev())
        with pytest.raises(_lib.MpxError):
            _lib.call("mpx_linear_ws", _lib.ptr(xw), xw.stride(0), _lib.ptr(w), _lib.ptr(b), M, N, K, 0, _lib.ptr(y0), N,
                      _lib.ptr(small), 16)


def test_linear_few_rows_streams_the_weights():
    """M <= 8 (the head of a single-problem rollout): rows in LDS, one wave per two output columns."""
    from mpinets_amd.pointnet2 import linear

    rng = np.random.default_rng(6)
    for (M, N, K) in [(1, 4096, 1024), (1, 2048, 4096), (1, 7, 128), (2, 513, 2112), (3, 64, 8), (4, 2048, 4096),
                      (5, 300, 2048), (8, 129, 260), (8, 33, 2048)]:
        xw = T(rng.normal(size=(M, K + 4)).astype(np.float32))
        w = T(rng.normal(size=(N, K)).astype(np.float32))
        b = T(rng.normal(size=N).astype(np.float32))
        out = torch.full((M, N + 5), -5.0, device=dev())
        for act, bias in ((0, b), (1, b), (2, None)):
            linear(xw[:, :K], w, bias, act, out=out[:, 2:2 + N])
            ref = xw[:, :K].double() @ w.double().T + (b.double() if bias is not None else 0)
            if act == 1:
                ref = ref.clamp(min=0)
            if act == 2:
                ref = torch.where(ref >= 0, ref, 0.01 * ref)
            assert (out[:, 2:2 + N].double() - ref).abs().max() <= 1e-5 * np.sqrt(K), (M, N, K, act)
            assert (out[:, :2] == -5).all() and (out[:, 2 + N:] == -5).all()


def test_groupnorm_leaky_and_rowmax(oracle):
    from mpinets_amd import _lib
    from mpinets_amd.pointnet2 import groupnorm_leaky

    rng = np.random.default_rng(2)
    for C in (4096, 2048, 64):
        x = (rng.normal(size=(9, C)) * 3 + 1).astype(np.float32)
        g, b = rng.normal(size=C).astype(np.float32), rng.normal(size=C).astype(np.float32)
        y = groupnorm_leaky(T(x), T(g), T(b), 16).cpu().numpy()
        ref = oracle._leaky(oracle._group_norm(x, 16, g, b))
        np.testing.assert_allclose(y, ref, rtol=0, atol=2e-5)
    x = rng.normal(size=(6 * 128, 300)).astype(np.float32)
    tx = T(x)
    y = torch.empty(6, 300, device=dev())
    _lib.call("mpx_rowmax", _lib.ptr(tx), 300, 6, 128, 300, _lib.ptr(y), 300)
    np.testing.assert_array_equal(y.cpu().numpy(), x.reshape(6, 128, 300).max(1))


def _sa_inputs(B, N, C, npoint, radius, seed):
    rng = np.random.default_rng(seed)
    xyz = (rng.uniform(-1, 1, (B, N, 3)) * 0.5).astype(np.float32)
    feat = rng.normal(size=(B, C, N)).astype(np.float32)
    return xyz, feat


@pytest.mark.parametrize("cfg", [dict(C=1, mlp=[1, 64, 64, 64], N=3000, npoint=64, radius=0.12),
                                 dict(C=64, mlp=[64, 128, 128, 256], N=512, npoint=32, radius=0.3)])
def test_sa_module_matches_oracle(oracle, cfg):
    """PointnetSAModule drop-in: (new_xyz, new_features) vs the oracle's unfused restatement."""
    from mpinets_amd.pointnet2 import PointnetSAModule

    torch.manual_seed(7)
    mod = PointnetSAModule(npoint=cfg["npoint"], radius=cfg["radius"], nsample=128, mlp=list(cfg["mlp"]), bn=False).to(dev())
    xyz, feat = _sa_inputs(2, cfg["N"], cfg["C"], cfg["npoint"], cfg["radius"], 3)
    with torch.no_grad():
        nx, nf = mod(T(xyz), T(feat))
    layers = [(c.weight.detach().cpu().numpy(), c.bias.detach().cpu().numpy()) for c in mod.convs()]
    onx, onf, _ = oracle.sa_module(xyz, feat, cfg["npoint"], cfg["radius"], 128, layers)
    np.testing.assert_array_equal(nx.cpu().numpy(), onx)
    assert nf.shape == (2, cfg["mlp"][-1], cfg["npoint"])
    np.testing.assert_allclose(nf.cpu().numpy(), onf, rtol=1e-5, atol=TOL)


def test_group_all_module_matches_oracle(oracle):
    from mpinets_amd.pointnet2 import PointnetSAModule

    torch.manual_seed(8)
    mod = PointnetSAModule(mlp=[256, 512, 512, 1024], bn=False).to(dev())
    xyz, feat = _sa_inputs(3, 128, 256, None, None, 4)
    with torch.no_grad():
        nx, nf = mod(T(xyz), T(feat))
    layers = [(c.weight.detach().cpu().numpy(), c.bias.detach().cpu().numpy()) for c in mod.convs()]
    _, onf, _ = oracle.sa_module(xyz, feat, None, None, None, layers)
    assert nx is None and nf.shape == (3, 1024, 1)
    np.testing.assert_allclose(nf.cpu().numpy(), onf, rtol=1e-5, atol=TOL)


def _state(mdl):
    return {k: v.detach().cpu().numpy() for k, v in mdl.state_dict().items()}


def test_policy_forward_matches_oracle(oracle):
    """model.forward(xyz, q): bit-exact FPS / ball-query indices, policy deltas within 1e-5."""
    from mpinets_amd.model import MotionPolicyNetwork
    from mpinets_amd.scenes import make_problem_batch

    torch.manual_seed(0)
    mdl = MotionPolicyNetwork().to(dev()).eval()
    prob = make_problem_batch(3, seed=1, device=dev(), kinds=("tabletop", "cubby", "dresser"), M1=40)
    aux = {}
    with torch.no_grad():
        dq = mdl(prob["xyz"], prob["q_norm"], aux=aux)
    odq, oaux = oracle.policy_forward(_state(mdl), prob["xyz"].cpu().numpy(), prob["q_norm"].cpu().numpy())
    np.testing.assert_array_equal(aux["fps_idx1"].cpu().numpy(), oaux["sa1"]["fps_idx"])
    np.testing.assert_array_equal(aux["ball_idx1"].cpu().numpy(), oaux["sa1"]["ball_idx"])
    np.testing.assert_array_equal(aux["fps_idx2"].cpu().numpy(), oaux["sa2"]["fps_idx"])
    np.testing.assert_array_equal(aux["ball_idx2"].cpu().numpy(), oaux["sa2"]["ball_idx"])
    np.testing.assert_allclose(aux["f1"].cpu().numpy(), oaux["f1"].transpose(0, 2, 1), rtol=1e-5, atol=TOL)
    np.testing.assert_allclose(aux["f3"].cpu().numpy(), oaux["f3"][:, :, 0], rtol=1e-5, atol=TOL)
    np.testing.assert_allclose(aux["encoding"].cpu().numpy(), oaux["encoding"], rtol=1e-4, atol=TOL)
    err = np.abs(dq.cpu().numpy() - odq).max()
    print("policy delta max abs err vs oracle: %.3e (|dq| max %.3e)" % (err, np.abs(odq).max()))
    assert err <= TOL
    # module-by-module path (reference shape conventions) gives the same encoding
    with torch.no_grad():
        enc2 = mdl.point_cloud_encoder.forward_modules(prob["xyz"])
    np.testing.assert_allclose(enc2.cpu().numpy(), aux["encoding"].cpu().numpy(), rtol=1e-5, atol=TOL)


def test_state_dict_keys_and_checkpoint_roundtrip(tmp_path):
    from mpinets_amd.model import MotionPolicyNetwork

    mdl = MotionPolicyNetwork()
    keys = set(mdl.state_dict().keys())
    for k in ("point_cloud_encoder.SA_modules.0.mlps.0.0.weight", "point_cloud_encoder.SA_modules.2.mlps.0.4.bias",
              "point_cloud_encoder.fc_layer.6.weight", "point_cloud_encoder.fc_layer.4.bias",
              "feature_encoder.8.weight", "decoder.6.bias"):
        assert k in keys
    assert mdl.state_dict()["point_cloud_encoder.SA_modules.0.mlps.0.0.weight"].shape == (64, 4, 1, 1)
    assert sum(p.numel() for p in mdl.parameters()) == 19068103
    path = tmp_path / "m.ckpt"
    torch.save({"state_dict": mdl.state_dict()}, path)
    m2 = MotionPolicyNetwork.load_from_checkpoint(str(path))
    for k, v in mdl.state_dict().items():
        assert torch.equal(v, m2.state_dict()[k])


def test_rollout_matches_oracle_and_mutates_slab(oracle):
    """TrainingMotionPolicyNetwork.rollout (model.py:128-183) for 3 steps vs an oracle loop."""
    from mpinets_amd import franka_tables as ft
    from mpinets_amd.model import TrainingMotionPolicyNetwork
    from mpinets_amd.robot import FrankaSampler
    from mpinets_amd.scenes import make_problem_batch

    torch.manual_seed(1)
    mdl = TrainingMotionPolicyNetwork(num_robot_points=2048).to(dev()).eval()
    prob = make_problem_batch(2, seed=2, device=dev())
    smp = FrankaSampler(dev())
    subset = prob["robot_subset"]

    def sampler(q):
        out = torch.empty((q.size(0), 2048, 3), device=dev())
        smp.sample_into(q, out, subset)
        return out

    xyz0 = prob["xyz"].clone()
    batch = {"xyz": prob["xyz"], "configuration": prob["q_norm"]}
    with torch.no_grad():
        traj = mdl.rollout(batch, 3, sampler, unnormalize=True)
    assert len(traj) == 4 and traj[0].shape == (2, 7)
    assert not torch.equal(prob["xyz"][:, :2048, :3], xyz0[:, :2048, :3])  # robot rows rewritten in place
    assert torch.equal(prob["xyz"][:, 2048:], xyz0[:, 2048:])
    # oracle loop
    sd = {k: v.detach().cpu().numpy() for k, v in mdl.state_dict().items()}
    x = xyz0.cpu().numpy().copy()
    q = prob["q_norm"].cpu().numpy()
    lim = ft.JOINT_LIMITS_REAL
    otraj = [oracle.unnormalize(q, lim)]
    for _ in range(3):
        dq, _ = oracle.policy_forward(sd, x, q)
        q = np.clip(q + dq, -1, 1).astype(np.float32)
        qu = oracle.unnormalize(q, lim)
        otraj.append(qu)
        x[:, :2048, :3] = oracle.transform_table(oracle.franka_fk(qu), smp.table_pts.cpu().numpy(),
                                                 smp.table_link.cpu().numpy(), subset.cpu().numpy())
    for a, b in zip(traj, otraj):
        np.testing.assert_allclose(a.cpu().numpy(), b, rtol=0, atol=5e-5)


# ---- split-bf16 ("bf16x3") fast mode: same interfaces, products on the bf16 matrix cores ---------------------------
@pytest.mark.parametrize("cfg", [dict(C=1, mlp=[1, 64, 64, 64], N=3000, npoint=70, radius=0.12),
                                 dict(C=64, mlp=[64, 128, 128, 256], N=512, npoint=37, radius=0.3)])
def test_sa_module_bf16x3_close_to_fp32_oracle(oracle, cfg):
    from mpinets_amd.pointnet2 import PointnetSAModule

    torch.manual_seed(7)
    mod = PointnetSAModule(npoint=cfg["npoint"], radius=cfg["radius"], nsample=128, mlp=list(cfg["mlp"]), bn=False,
                           precision="bf16x3").to(dev())
    xyz, feat = _sa_inputs(2, cfg["N"], cfg["C"], cfg["npoint"], cfg["radius"], 3)
    with torch.no_grad():
        nx, nf = mod(T(xyz), T(feat))
        mod.precision = "fp32"
        _, nf32 = mod(T(xyz), T(feat))
    layers = [(c.weight.detach().cpu().numpy(), c.bias.detach().cpu().numpy()) for c in mod.convs()]
    onx, onf, _ = oracle.sa_module(xyz, feat, cfg["npoint"], cfg["radius"], 128, layers)
    np.testing.assert_array_equal(nx.cpu().numpy(), onx)  # indices / centres never depend on the MLP precision
    err = np.abs(nf.cpu().numpy() - onf).max()
    print("bf16x3 SA features: max abs err %.3e (fp32 kernel: %.3e), |f| max %.2f" % (
        err, np.abs(nf32.cpu().numpy() - onf).max(), np.abs(onf).max()))
    np.testing.assert_allclose(nf.cpu().numpy(), onf, rtol=1e-4, atol=1e-4)


def test_policy_forward_bf16x3_within_tolerance(oracle):
    """The fast mode must still meet the north-star bar: bit-exact indices, policy deltas within 1e-5."""
    from mpinets_amd.model import MotionPolicyNetwork
    from mpinets_amd.scenes import make_problem_batch

    torch.manual_seed(0)
    mdl = MotionPolicyNetwork().to(dev()).eval().set_precision("bf16x3")
    prob = make_problem_batch(3, seed=1, device=dev(), kinds=("tabletop", "cubby", "dresser"), M1=40)
    aux = {}
    with torch.no_grad():
        dq = mdl(prob["xyz"], prob["q_norm"], aux=aux)
        dq32 = mdl.set_precision("fp32")(prob["xyz"], prob["q_norm"])
    odq, oaux = oracle.policy_forward(_state(mdl), prob["xyz"].cpu().numpy(), prob["q_norm"].cpu().numpy())
    np.testing.assert_array_equal(aux["fps_idx2"].cpu().numpy(), oaux["sa2"]["fps_idx"])
    np.testing.assert_array_equal(aux["ball_idx2"].cpu().numpy(), oaux["sa2"]["ball_idx"])
    err = np.abs(dq.cpu().numpy() - odq).max()
    print("bf16x3 policy delta max abs err vs fp32 oracle: %.3e (fp32 kernels: %.3e)" % (
        err, np.abs(dq32.cpu().numpy() - odq).max()))
    assert err <= TOL


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_padding_elision_is_bit_identical(precision):
    """Skipping neighbourhood tiles that hold only ball-query padding must not change a single bit."""
    from mpinets_amd.model import MotionPolicyNetwork
    from mpinets_amd.scenes import make_problem_batch

    torch.manual_seed(5)
    mdl = MotionPolicyNetwork().to(dev()).eval().set_precision(precision)
    prob = make_problem_batch(5, seed=9, device=dev(), kinds=("tabletop", "cubby", "dresser"), M1=40, device_clouds=True)
    a, b = {}, {}
    with torch.no_grad():
        dq_on = mdl.set_elide_padding(True)(prob["xyz"], prob["q_norm"], aux=a)
        dq_off = mdl.set_elide_padding(False)(prob["xyz"], prob["q_norm"], aux=b)
    assert torch.equal(a["f1"], b["f1"]) and torch.equal(a["sa3_in"], b["sa3_in"]) and torch.equal(dq_on, dq_off)
    c1, c2 = a["ball_cnt1"].float(), a["ball_cnt2"].float()
    print("mean distinct neighbours: SA1 %.1f, SA2 %.1f of 128" % (c1.mean(), c2.mean()))
    assert c1.min() >= 1 and c2.max() <= 128


@pytest.mark.parametrize("precision,factored", [("fp32", True), ("fp32", False), ("bf16x3", True), ("bf16x3", False)])
def test_hits_only_ball_query_rows_do_not_change_the_forward(precision, factored):
    """Without ``aux`` the forward's ball queries write the hit slots only (mpx_ball_query_hits); with ``aux`` they pad the
    rows the way pointnet2_ops does.  The grouped-MLP kernels never read past the counts: same output bits, also when the
    index rows held a wrong index in every slot before the call (a poisoned caching-allocator block)."""
    from mpinets_amd.model import MotionPolicyNetwork
    from mpinets_amd.scenes import make_problem_batch

    torch.manual_seed(8)
    mdl = MotionPolicyNetwork().to(dev()).eval().set_precision(precision).set_factored(factored)
    prob = make_problem_batch(6, seed=21, device=dev(), kinds=("tabletop", "cubby", "dresser"), M1=40, device_clouds=True)
    with torch.no_grad():
        padded = mdl(prob["xyz"], prob["q_norm"], aux={})
        torch.cuda.synchronize()
        for _ in range(3):  # blocks of the sizes the forward takes, released full of a valid but wrong point index
            junk = [torch.full((6, n, 128), 5, dtype=torch.int32, device=dev()) for n in (512, 128)]
            del junk
            hits = mdl(prob["xyz"], prob["q_norm"])
            assert torch.equal(padded, hits)


@pytest.mark.parametrize("factored", [True, False])
def test_small_batch_launch_shape_is_bit_identical(factored):
    """A single problem runs the grouped-MLP kernels with fewer queries per wave than a large batch (more waves,
    lower latency): every (query, neighbour) row is still the same arithmetic, so the features agree bit for bit."""
    from mpinets_amd.model import MotionPolicyNetwork
    from mpinets_amd.scenes import make_problem_batch

    torch.manual_seed(6)
    mdl = MotionPolicyNetwork().to(dev()).eval().set_factored(factored)
    prob = make_problem_batch(80, seed=12, device=dev(), kinds=("tabletop", "cubby", "dresser"), M1=40, device_clouds=True)
    a, b = {}, {}
    with torch.no_grad():
        mdl(prob["xyz"], prob["q_norm"], aux=a)                      # 80 x 512 / 80 x 128 queries: 16 / 8 per wave
        dq_small = mdl(prob["xyz"][:3].contiguous(), prob["q_norm"][:3].contiguous(), aux=b)  # 4 / 2 per wave
    assert torch.equal(a["f1"][:3], b["f1"]) and torch.equal(a["sa3_in"][:3], b["sa3_in"])
    assert torch.isfinite(dq_small).all()


def test_factored_first_layer_matches_direct_form(oracle):
    """SA2's first layer per point / per query (mpx_sa_mlp_factored) vs per (query, neighbour) row (mpx_sa_mlp):
    same result up to the rounding of one re-associated sum, both within the north-star 1e-5 of the oracle."""
    from mpinets_amd.model import MotionPolicyNetwork
    from mpinets_amd.pointnet2 import PointnetSAModule
    from mpinets_amd.scenes import make_problem_batch

    torch.manual_seed(7)
    mdl = MotionPolicyNetwork().to(dev()).eval()
    prob = make_problem_batch(3, seed=4, device=dev(), kinds=("tabletop", "dresser"), M1=40, device_clouds=True)
    a, b = {}, {}
    with torch.no_grad():
        dq_f = mdl.set_factored(True)(prob["xyz"], prob["q_norm"], aux=a)
        dq_d = mdl.set_factored(False)(prob["xyz"], prob["q_norm"], aux=b)
    mdl.set_factored(True)
    f2f, f2d = a["sa3_in"][:, :, 3:259], b["sa3_in"][:, :, 3:259]
    assert not torch.equal(f2f, f2d)  # two different kernels really ran
    assert (f2f - f2d).abs().max().item() <= 2e-6 * f2d.abs().max().item()
    sd = {k: v.detach().cpu().numpy() for k, v in mdl.state_dict().items()}
    ref, _ = oracle.policy_forward(sd, prob["xyz"].cpu().numpy(), prob["q_norm"].cpu().numpy())
    assert np.abs(dq_f.cpu().numpy() - ref).max() < 1e-5 and np.abs(dq_d.cpu().numpy() - ref).max() < 1e-5
    print("factored vs oracle %.2e, direct vs oracle %.2e" % (np.abs(dq_f.cpu().numpy() - ref).max(),
                                                            np.abs(dq_d.cpu().numpy() - ref).max()))
    # the module-level API (reference shapes) takes the same route
    sa2 = mdl.point_cloud_encoder.SA_modules[1]
    xyz1, f1 = a["xyz1"].contiguous(), a["f1"].transpose(1, 2).contiguous()
    with torch.no_grad():
        _, o_f = sa2(xyz1, f1)
        sa2.factored = False
        _, o_d = sa2(xyz1, f1)
        sa2.factored = True
    assert (o_f - o_d).abs().max().item() <= 2e-6 * o_d.abs().max().item()
    np.testing.assert_allclose(o_f.transpose(1, 2).cpu().numpy(), f2f.cpu().numpy(), rtol=0, atol=1e-6)


def test_linear_rowmax_equals_linear_then_rowmax():
    from mpinets_amd import _lib
    from mpinets_amd.pointnet2 import linear

    rng = np.random.default_rng(4)
    x = T(rng.normal(size=(5 * 128, 512)).astype(np.float32))
    w = T((rng.normal(size=(1024, 512)) * 0.05).astype(np.float32))
    b = T(rng.normal(size=1024).astype(np.float32))
    full = torch.empty((640, 1024), device=dev())  # the unsplit kernel: same k order as the pooled one
    _lib.call("mpx_linear", _lib.ptr(x), 512, _lib.ptr(w), _lib.ptr(b), 640, 1024, 512, 1, _lib.ptr(full), 1024)
    ref = full.reshape(5, 128, 1024).max(dim=1).values
    assert (linear(x, w, b, 1) - full).abs().max() < 1e-5  # (linear() splits K at this size)
    out = torch.full((5, 1024), -1.0, device=dev())
    _lib.call("mpx_linear_rowmax", _lib.ptr(x), 512, _lib.ptr(w), _lib.ptr(b), 640, 1024, 512, 128, _lib.ptr(out), 1024)
    assert torch.equal(out, ref)


def test_linear_bf16x3_matches_fp32_layer():
    """Split-bf16 dense layer vs float64: three products per fp32 product keep ~2^-16 relative accuracy per term."""
    from mpinets_amd import _lib
    from mpinets_amd.pointnet2 import SplitWeights, linear, linear_x3

    rng = np.random.default_rng(11)
    split = SplitWeights()
    for (M, N, K) in [(128, 128, 16), (257, 200, 260), (5, 4096, 1024), (1000, 64, 2112), (640, 1024, 512)]:
        x = rng.normal(size=(M, K)).astype(np.float32)
        w = (rng.normal(size=(N, K)) / np.sqrt(K)).astype(np.float32)
        b = rng.normal(size=N).astype(np.float32)
        ref = x.astype(np.float64) @ w.astype(np.float64).T + b
        for act in (0, 1, 2):
            r = ref if act == 0 else (np.maximum(ref, 0) if act == 1 else np.where(ref >= 0, ref, 0.01 * ref))
            y = linear_x3(T(x), T(w), T(b), act, split).cpu().numpy()
            y32 = linear(T(x), T(w), T(b), act).cpu().numpy()
            e3, e32 = np.abs(y - r).max(), np.abs(y32 - r).max()
            # a split operand carries ~2^-17 relative error (x_hi + x_lo != x exactly) and x_lo*w_lo is dropped:
            # ~1e-5 of sum |x||w| (here a few units), against ~1e-6 for the fp32 kernel
            assert e3 <= 1e-4 and e32 <= 2e-5, (M, N, K, act, e3, e32)
    # pooled variant == linear_x3 + ReLU + max over each 128-row group
    x = T(rng.normal(size=(5 * 128, 512)).astype(np.float32))
    w = T((rng.normal(size=(1024, 512)) * 0.05).astype(np.float32))
    b = T(rng.normal(size=1024).astype(np.float32))
    out = torch.full((5, 1024), -1.0, device=dev())
    _lib.call("mpx_linear_rowmax_bf16x3", _lib.ptr(x), 512, _lib.ptr(split.get(w)), _lib.ptr(b), 640, 1024, 512, 128,
              _lib.ptr(out), 1024)
    ref = linear_x3(x, w, b, 1, split).reshape(5, 128, 1024).max(dim=1).values
    assert torch.equal(out, ref)
    # a changed parameter refreshes the planes
    w.mul_(2.0)
    y2 = linear_x3(x, w, b, 0, split)
    np.testing.assert_allclose(y2.cpu().numpy(), (x.double() @ w.double().T + b.double()).cpu().numpy(), atol=3e-4)


def _pairs_of(x):
    """torch restatement of the pairs form: per 16 k-values [hi x 16 | lo x 16], hi = bf16(v), lo = bf16(v - hi)."""
    R, K = x.shape
    Kp = (K + 15) // 16 * 16
    xp = torch.nn.functional.pad(x, (0, Kp - K))
    hi = xp.to(torch.bfloat16)
    lo = (xp - hi.float()).to(torch.bfloat16)
    return torch.cat([hi.view(R, Kp // 16, 16), lo.view(R, Kp // 16, 16)], dim=2).reshape(R, 2 * Kp).contiguous()


def test_linear_bf16x3_pairs_chain_is_bit_identical_to_the_row_chain():
    """The pairs-input / pairs-output forms of the bf16x3 dense layer (operands already split by the layer before,
    staged by DMA) == the fp32-row form bit for bit: fp32 rows out, pairs out, and the pooled variants; ragged M / N,
    every activation; and the three-layer group-all chain through pairs == the chain through fp32 rows."""
    from mpinets_amd import _lib
    from mpinets_amd.pointnet2 import SplitWeights, linear_x3, split_pairs

    rng = np.random.default_rng(23)
    split = SplitWeights()
    for (M, N, K) in [(128, 128, 16), (257, 200, 48), (5, 4096, 1024), (1000, 64, 2112), (640, 1024, 512), (1, 4, 32),
                      (300, 36, 20)]:
        x = T(rng.normal(size=(M, K)).astype(np.float32))
        w = T((rng.normal(size=(N, K)) / np.sqrt(K)).astype(np.float32))
        b = T(rng.normal(size=N).astype(np.float32))
        wp = split.get(w)
        assert torch.equal(wp, _pairs_of(w)) and torch.equal(split_pairs(x), _pairs_of(x))  # mpx_split_bf16 itself
        Np = (N + 15) // 16 * 16
        for act in (0, 1, 2):
            ref = linear_x3(x, w, b, act, split)
            rp = _pairs_of(ref)
            # fp32 rows in -> pairs out (the pad columns of the last 16-group are left to the caller: pre-zeroed here)
            yp = torch.zeros((M, 2 * Np), dtype=torch.bfloat16, device=dev())
            _lib.call("mpx_linear_bf16x3_to_pairs", _lib.ptr(x), K, _lib.ptr(wp), _lib.ptr(b), M, N, K, act, _lib.ptr(yp), 2 * Np)
            assert torch.equal(yp, rp), ("to_pairs", M, N, K, act)
            if K % 16:
                continue
            xp = split_pairs(x)
            y = torch.full((M, N), 7.0, device=dev())
            _lib.call("mpx_linear_bf16x3_pairs", _lib.ptr(xp), 2 * K, _lib.ptr(wp), _lib.ptr(b), M, N, K, act, _lib.ptr(y), N,
                      None, 0)
            assert torch.equal(y, ref), ("pairs", M, N, K, act)
            yp.zero_()
            _lib.call("mpx_linear_bf16x3_pairs", _lib.ptr(xp), 2 * K, _lib.ptr(wp), _lib.ptr(b), M, N, K, act, None, 0,
                      _lib.ptr(yp), 2 * Np)
            assert torch.equal(yp, rp), ("pairs->pairs", M, N, K, act)
    # padded leading dimensions: activation pairs with lda > 2 K, output pairs with ldp > 2 Np (columns beyond untouched)
    M, N, K = 300, 96, 64
    x = T(rng.normal(size=(M, K)).astype(np.float32))
    w = T(rng.normal(size=(N, K)).astype(np.float32))
    xp = torch.nn.functional.pad(split_pairs(x), (0, 8)).contiguous()
    yp = torch.full((M, 2 * N + 4), 3.0, dtype=torch.bfloat16, device=dev())
    _lib.call("mpx_linear_bf16x3_pairs", _lib.ptr(xp), 2 * K + 8, _lib.ptr(split.get(w)), None, M, N, K, 0, None, 0, _lib.ptr(yp),
              2 * N + 4)
    assert torch.equal(yp[:, :2 * N], _pairs_of(linear_x3(x, w, None, 0, split))) and bool((yp[:, 2 * N:] == 3).all())
    # the group-all chain: 272 -> 512 -> 512 -> 1024 + max over 128 rows, 5 environments (an odd number of 128-row
    # groups: the last 256-row tile is half empty), pooled as fp32 and as pairs
    x = T(np.maximum(rng.normal(size=(640, 272)), 0).astype(np.float32))
    ws = [T((rng.normal(size=s) * 0.05).astype(np.float32)) for s in ((512, 272), (512, 512), (1024, 512))]
    bs = [T(rng.normal(size=n).astype(np.float32)) for n in (512, 512, 1024)]
    h = linear_x3(linear_x3(x, ws[0], bs[0], 1, split), ws[1], bs[1], 1, split)
    ref = torch.empty((5, 1024), device=dev())
    _lib.call("mpx_linear_rowmax_bf16x3", _lib.ptr(h), 512, _lib.ptr(split.get(ws[2])), _lib.ptr(bs[2]), 640, 1024, 512, 128,
              _lib.ptr(ref), 1024)
    p1 = torch.empty((640, 1024), dtype=torch.bfloat16, device=dev())
    p2 = torch.empty_like(p1)
    _lib.call("mpx_linear_bf16x3_to_pairs", _lib.ptr(x), 272, _lib.ptr(split.get(ws[0])), _lib.ptr(bs[0]), 640, 512, 272, 1,
              _lib.ptr(p1), 1024)
    _lib.call("mpx_linear_bf16x3_pairs", _lib.ptr(p1), 1024, _lib.ptr(split.get(ws[1])), _lib.ptr(bs[1]), 640, 512, 512, 1, None,
              0, _lib.ptr(p2), 1024)
    out = torch.full((5, 1024), -1.0, device=dev())
    _lib.call("mpx_linear_rowmax_bf16x3_pairs", _lib.ptr(p2), 1024, _lib.ptr(split.get(ws[2])), _lib.ptr(bs[2]), 640, 1024, 512,
              128, _lib.ptr(out), 1024, None, 0)
    assert torch.equal(out, ref)
    outp = torch.zeros((5, 2048), dtype=torch.bfloat16, device=dev())
    _lib.call("mpx_linear_rowmax_bf16x3_pairs", _lib.ptr(p2), 1024, _lib.ptr(split.get(ws[2])), _lib.ptr(bs[2]), 640, 1024, 512,
              128, None, 0, _lib.ptr(outp), 2048)
    assert torch.equal(outp, _pairs_of(ref))
    # GroupNorm + LeakyReLU writing the next layer's pairs == its fp32 result, split
    from mpinets_amd.pointnet2 import groupnorm_leaky
    xg = T(rng.normal(size=(37, 4096)).astype(np.float32))
    gam, bet = T(rng.normal(size=4096).astype(np.float32)), T(rng.normal(size=4096).astype(np.float32))
    gp = torch.zeros((37, 8192), dtype=torch.bfloat16, device=dev())
    _lib.call("mpx_groupnorm_leaky_to_pairs", _lib.ptr(xg), _lib.ptr(gam), _lib.ptr(bet), 37, 4096, 16, 1e-5, _lib.ptr(gp), 8192)
    assert torch.equal(gp, _pairs_of(groupnorm_leaky(xg, gam, bet, 16)))
    # argument checks: K must be whole 16-k groups, exactly one output form
    with pytest.raises(_lib.MpxError):
        _lib.call("mpx_linear_bf16x3_pairs", _lib.ptr(p1), 1024, _lib.ptr(split.get(ws[1])), None, 640, 512, 40, 0, None, 0,
                  _lib.ptr(p2), 1024)
    with pytest.raises(_lib.MpxError):
        _lib.call("mpx_linear_bf16x3_pairs", _lib.ptr(p1), 1024, _lib.ptr(split.get(ws[1])), None, 640, 512, 512, 0,
                  _lib.ptr(out), 512, _lib.ptr(p2), 1024)


def test_policy_forward_bf16x3_pairs_on_and_off_agree():
    """The policy forward in bf16x3 with the group-all MLP layer by layer through pairs == with fp32 rows, bit for bit; the
    default -- its first two layers as ONE kernel (mpx_sa3_front_bf16x3: another order of the 16 products inside an MFMA
    step, the bias added last instead of in the epilogue's place) -- equals both to rounding."""
    from mpinets_amd.model import MotionPolicyNetwork
    from mpinets_amd.scenes import make_problem_batch

    torch.manual_seed(5)
    mdl = MotionPolicyNetwork().to(dev()).eval().set_precision("bf16x3")
    enc = mdl.point_cloud_encoder
    prob = make_problem_batch(6, seed=31, device=dev())
    with torch.no_grad():
        fused = mdl(prob["xyz"], prob["q_norm"]).clone()
        assert enc._sa3_fp is not None  # (the fused kernel served the default call)
        enc.sa3_front_fused = False
        a = mdl(prob["xyz"], prob["q_norm"]).clone()
        enc.dense_through_pairs = False
        b = mdl(prob["xyz"], prob["q_norm"]).clone()
    assert torch.equal(a, b)
    err = (fused - a).abs().max().item()
    print("bf16x3: fused group-all front vs layer by layer: %.2e" % err)
    assert err <= 2e-6


@pytest.mark.parametrize("B", [1, 5, 9, 40])
def test_single_call_c_forward_reproduces_the_python_path(oracle, B):
    """mpx_policy_forward (one C call, caller workspace, no Python orchestration) == MotionPolicyNetwork.forward
    bit for bit -- across the batch sizes where the launch shapes change (weight-streaming layers, split-K,
    queries per wave, fused / unfused pooling) -- and, like it, within 1e-5 of the oracle."""
    from mpinets_amd import _lib
    from mpinets_amd.model import MotionPolicyNetwork
    from mpinets_amd.scenes import make_problem_batch

    torch.manual_seed(11)
    mdl = MotionPolicyNetwork().to(dev()).eval()
    prob = make_problem_batch(B, seed=21, device=dev(), kinds=("tabletop", "cubby", "dresser"), M1=40, device_clouds=True)
    with torch.no_grad():
        ref = mdl(prob["xyz"], prob["q_norm"])
        got = mdl.forward_native(prob["xyz"], prob["q_norm"])
        again = mdl.forward_native(prob["xyz"], prob["q_norm"])
    assert torch.equal(got, ref) and torch.equal(again, got)
    if B <= 5:
        sd = {k: v.detach().cpu().numpy() for k, v in mdl.state_dict().items()}
        want, _ = oracle.policy_forward(sd, prob["xyz"].cpu().numpy(), prob["q_norm"].cpu().numpy())
        assert np.abs(got.cpu().numpy() - want).max() < TOL
    # argument checking: a workspace that is too small is refused with a message, nothing is launched
    w, keep = mdl.native_weights()
    need = _lib.load().mpx_policy_workspace(B, prob["xyz"].size(1))
    small = torch.empty(need - 256, dtype=torch.uint8, device=dev())
    dq = torch.empty((B, 7), device=dev())
    import ctypes
    with pytest.raises(_lib.MpxError, match="workspace"):
        _lib.call("mpx_policy_forward", ctypes.addressof(w), _lib.ptr(prob["xyz"]), prob["xyz"].size(1), _lib.ptr(prob["q_norm"]),
                  B, _lib.ptr(dq), _lib.ptr(small), need - 256)


@pytest.mark.parametrize("B", [1, 24])
def test_single_call_rollout_step_reproduces_the_engine(B):
    """mpx_rollout_step (policy forward + joint update + FK cloud refresh + collision check in one C call) leaves
    exactly the state RolloutEngine.step() leaves: joint angles, normalised angles, slab and collision flags."""
    from mpinets_amd.model import MotionPolicyNetwork
    from mpinets_amd.rollout import RolloutEngine
    from mpinets_amd.scenes import make_problem_batch

    torch.manual_seed(12)
    mdl = MotionPolicyNetwork().to(dev()).eval()
    engines = []
    for _ in range(2):
        prob = make_problem_batch(B, seed=31, device=dev(), kinds=("tabletop", "cubby"), M1=24, device_clouds=True)
        engines.append(RolloutEngine(mdl, prob))
    a, b = engines
    assert torch.equal(a.xyz, b.xyz)
    for _ in range(3):
        qa, qb = a.step(), b.step_native()
        assert torch.equal(qa, qb) and torch.equal(a.q_norm, b.q_norm)
        assert torch.equal(a.xyz, b.xyz) and torch.equal(a.flags, b.flags)
    assert (a.q_norm.abs() <= 1).all()


@pytest.mark.parametrize("B", [1, 24])
def test_native_rollout_with_rerender_and_success_tracking_reproduces_the_engine(B):
    """mpx_rollout -- several closed-loop steps in ONE C call with the per-step scene re-render (seed schedule, global
    environment ids) and the on-device early-stop bookkeeping of rollout_until_success (run_inference.py:171-189) --
    leaves exactly the state the Python engine leaves: joint angles, slab (scene rows included), flags, done flags,
    per-environment step counts and the trajectory rows."""
    from mpinets_amd import franka_tables as ft
    from mpinets_amd.model import MotionPolicyNetwork
    from mpinets_amd.robot import franka_fk, frames_to_matrix
    from mpinets_amd.rollout import RolloutEngine
    from mpinets_amd.scenes import make_problem_batch

    torch.manual_seed(12)
    mdl = MotionPolicyNetwork().to(dev()).eval()
    engines, targets = [], None
    for _ in range(2):
        prob = make_problem_batch(B, seed=31, device=dev(), kinds=("tabletop", "cubby", "dresser"), M1=40, M2=16,
                                  device_clouds=True, env_offset=3, total_envs=3 + B)
        eng = RolloutEngine(mdl, prob, rerender_scene=True, scene_seed=77)
        if targets is None:  # environment 0 starts at its target (done after the first step's test unless it moves away)
            targets = prob["target_pose"].clone()
            targets[0] = frames_to_matrix(franka_fk(prob["q"])[:, ft.LINK_ID["right_gripper"]])[0]
        eng.track_success(targets, pos_tol=0.5, rot_tol_deg=120.0)  # loose: some environments finish within 4 steps
        engines.append(eng)
    a, b = engines
    L = 4
    traj_a = [a.step().clone() for _ in range(L)]
    traj_b = torch.zeros((B, L + 1, 7), device=dev())
    b.run_native(1, trajectory=traj_b, trajectory_row=1)  # one step, then three in one call (first_step continues)
    b.run_native(L - 1, trajectory=traj_b, trajectory_row=2)
    assert a.steps_done == b.steps_done == L
    assert torch.equal(a.q, b.q) and torch.equal(a.q_norm, b.q_norm)
    assert torch.equal(a.xyz, b.xyz), "slab (robot + re-rendered scene rows) differs"
    assert torch.equal(a.flags, b.flags) and torch.equal(a.done, b.done) and torch.equal(a.steps, b.steps)
    assert torch.equal(torch.stack(traj_a, 1), traj_b[:, 1:])
    assert (traj_b[:, 0] == 0).all()
    if B > 1:
        assert 0 < int((a.done != 0).sum()) < B  # the freeze path and the moving path were both exercised


def test_native_rollout_until_success_equals_python_loop():
    from mpinets_amd.model import MotionPolicyNetwork
    from mpinets_amd.rollout import RolloutEngine
    from mpinets_amd.scenes import make_problem_batch

    torch.manual_seed(5)
    mdl = MotionPolicyNetwork().to(dev()).eval()
    out = []
    for native in (False, True):
        prob = make_problem_batch(6, seed=9, device=dev(), kinds=("tabletop", "cubby"), M1=24, device_clouds=True)
        eng = RolloutEngine(mdl, prob)
        eng.track_success(prob["target_pose"], pos_tol=0.6, rot_tol_deg=150.0)
        out.append(eng.rollout_until_success(max_steps=7, check_every=3, native=native) + (eng.xyz, eng.flags))
    (ta, la, xa, fa), (tb, lb, xb, fb) = out
    assert ta.shape == tb.shape and torch.equal(ta, tb) and torch.equal(la, lb) and torch.equal(xa, xb) and torch.equal(fa, fb)


def test_policy_forward_is_safe_from_two_host_threads():
    """Two host threads drive mpx_policy_forward on the same device from different streams (small batches: the
    internal side stream + fork / join events are shared per device and must be held exclusively per call)."""
    import threading

    from mpinets_amd.model import MotionPolicyNetwork
    from mpinets_amd.scenes import make_problem_batch

    torch.manual_seed(2)
    mdl = MotionPolicyNetwork().to(dev()).eval()
    probs = [make_problem_batch(3, seed=40 + i, device=dev(), device_clouds=True) for i in range(2)]
    with torch.no_grad():
        want = [mdl.forward_native(p["xyz"], p["q_norm"]).clone() for p in probs]
    torch.cuda.synchronize()
    got, errs = [[], []], []

    def work(i):
        try:
            torch.cuda.set_device(dev())
            with torch.cuda.stream(torch.cuda.Stream(device=dev())), torch.no_grad():
                for _ in range(40):
                    got[i].append(mdl.forward_native(probs[i]["xyz"], probs[i]["q_norm"]))
                torch.cuda.current_stream().synchronize()
        except Exception as e:  # pragma: no cover
            errs.append(e)

    mdl.native_weights()  # (pack once, outside the threads)
    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    torch.cuda.synchronize()
    for i in range(2):
        assert all(torch.equal(g, want[i]) for g in got[i])


def test_lightning_style_checkpoint_and_cache_invalidation(tmp_path):
    """A checkpoint that pickles non-tensor objects (Lightning callbacks / hyper-parameters) loads; writes through
    ``param.data`` need ``invalidate_caches()`` (they do not bump the parameter version)."""
    import argparse

    from mpinets_amd.model import MotionPolicyNetwork
    from mpinets_amd.scenes import make_problem_batch

    torch.manual_seed(8)
    mdl = MotionPolicyNetwork()
    path = tmp_path / "lightning.ckpt"
    torch.save({"state_dict": mdl.state_dict(), "epoch": 3, "hyper_parameters": argparse.Namespace(lr=1e-4),
                "callbacks": {"ModelCheckpoint": {"best": argparse.Namespace(score=0.1)}}}, path)
    import pickle

    with pytest.raises(pickle.UnpicklingError, match="trust_checkpoint=True"):  # pickled objects: refused unless opted in
        MotionPolicyNetwork.load_from_checkpoint(str(path))
    with pytest.raises(Exception) as ei:  # a file that is simply not there is never retried unsafely
        MotionPolicyNetwork.load_from_checkpoint(str(path) + ".missing", trust_checkpoint=True)
    assert not isinstance(ei.value, pickle.UnpicklingError)
    with pytest.warns(UserWarning, match="weights_only=False"):
        m2 = MotionPolicyNetwork.load_from_checkpoint(str(path), trust_checkpoint=True).to(dev()).eval()
    for k, v in mdl.state_dict().items():
        assert torch.equal(v, m2.state_dict()[k].cpu())
    prob = make_problem_batch(2, seed=3, device=dev())
    with torch.no_grad():
        a = m2(prob["xyz"], prob["q_norm"]).clone()
        conv = m2.point_cloud_encoder.SA_modules[1].convs()[1]
        conv.weight.data.mul_(1.5)  # behind torch's back: the packed weights are stale ...
        stale = m2(prob["xyz"], prob["q_norm"]).clone()
        fresh = m2.invalidate_caches()(prob["xyz"], prob["q_norm"]).clone()  # ... until the caches are dropped
    assert torch.equal(stale, a) and not torch.equal(fresh, a)


def test_operands_on_another_device_are_rejected():
    from mpinets_amd import _lib

    if torch.cuda.device_count() < 2:
        class FakeOther:  # a stand-in with the attributes require_cuda looks at
            is_cuda = True
            device = torch.device("cuda", torch.cuda.current_device() + 1)
        with pytest.raises(_lib.MpxError, match="current device"):
            _lib.require_cuda(FakeOther())
    else:
        with pytest.raises(_lib.MpxError, match="current device"):
            _lib.require_cuda(torch.zeros(1, device="cuda:1"))


def test_group_all_chain_kernel_matches_float64_and_the_layered_path():
    """mpx_sa3_chain (the group-all module as ONE kernel: activations in LDS, nothing between the input rows and the pooled
    row in HBM) vs a float64 evaluation of the same three layers + max, and vs the layer-by-layer kernels it replaces
    (mpx_linear x 2 + mpx_linear_rowmax): same function, different fp32 summation order."""
    from mpinets_amd import _lib
    from mpinets_amd.model import MotionPolicyNetwork

    torch.manual_seed(12)
    mdl = MotionPolicyNetwork().to(dev()).eval()
    enc = mdl.point_cloud_encoder
    c3 = enc.SA_modules[2].convs()
    B, K3 = 300, 272
    x = torch.zeros((B * 128, K3), device=dev())
    x[:, :259] = torch.randn((B * 128, 259), device=dev()) * 0.7
    x[5 * 128:6 * 128] = 0.0  # an all-zero environment: the pooled row is relu of the bias chain
    pack = enc._sa3_pack(K3)
    assert pack is not None and pack.numel() == _lib.load().mpx_sa3_pack_size(272, 512, 512, 1024)
    out = torch.empty((B, 1024), device=dev())
    _lib.call("mpx_sa3_chain", _lib.ptr(x), K3, B, 128, _lib.ptr(pack), K3, 512, 512, 1024, _lib.ptr(out), 1024)
    w = [c.weight.detach().view(c.out_channels, -1).double() for c in c3]
    b = [c.bias.detach().double() for c in c3]
    h = torch.relu(x[:, :259].double() @ w[0].T + b[0])
    h = torch.relu(h @ w[1].T + b[1])
    h = torch.relu(h @ w[2].T + b[2])
    ref = h.view(B, 128, 1024).max(dim=1).values
    err = (out.double() - ref).abs().max().item()
    assert err < 1e-5 * max(1.0, ref.abs().max().item()), err
    # the layered kernels on the same rows
    from mpinets_amd.pointnet2 import linear

    h1 = linear(x, enc._sa3_first_weight(), c3[0].bias, 1)
    h2 = linear(h1, c3[1].weight.view(512, -1), c3[1].bias, 1)
    lay = torch.empty((B, 1024), device=dev())
    _lib.call("mpx_linear_rowmax", _lib.ptr(h2), 512, _lib.ptr(c3[2].weight.view(1024, -1)), _lib.ptr(c3[2].bias), B * 128, 1024,
              512, 128, _lib.ptr(lay), 1024)
    assert (out - lay).abs().max().item() < 2e-6 * max(1.0, ref.abs().max().item())
    # strided output rows, a second call is bit-identical, bad shapes are refused
    wide = torch.zeros((B, 1100), device=dev())
    _lib.call("mpx_sa3_chain", _lib.ptr(x), K3, B, 128, _lib.ptr(pack), K3, 512, 512, 1024, _lib.ptr(wide), 1100)
    assert torch.equal(wide[:, :1024], out) and (wide[:, 1024:] == 0).all()
    lib = _lib.load()
    assert lib.mpx_sa3_chain(x.data_ptr(), K3, B, 64, pack.data_ptr(), K3, 512, 512, 1024, out.data_ptr(), 1024, None) != 0
    assert lib.mpx_sa3_pack_size(272, 512, 512, 512) == -1


def test_policy_forward_is_consistent_across_the_chain_threshold():
    """B >= 256 problems take the fused group-all kernel, fewer the layered GEMMs: the same problems evaluated on either
    side of the threshold agree to fp32 rounding (and the single-call C forward makes the same choice: bit-identical)."""
    from mpinets_amd.model import SA3_CHAIN_MIN_BATCH, MotionPolicyNetwork
    from mpinets_amd.scenes import make_problem_batch

    torch.manual_seed(6)
    mdl = MotionPolicyNetwork().to(dev()).eval()
    B = SA3_CHAIN_MIN_BATCH
    prob = make_problem_batch(B, seed=41, device=dev(), kinds=("tabletop", "cubby", "dresser"), M1=40, scene_pool=32,
                              device_clouds=True)
    with torch.no_grad():
        big = mdl(prob["xyz"], prob["q_norm"]).clone()
        nat = mdl.forward_native(prob["xyz"], prob["q_norm"]).clone()
        lo = torch.cat([mdl(prob["xyz"][i:i + B // 2].contiguous(), prob["q_norm"][i:i + B // 2].contiguous())
                        for i in (0, B // 2)])
    assert torch.equal(big, nat)
    assert (big - lo).abs().max().item() < 2e-6, (big - lo).abs().max().item()


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_encoder_slabs_are_bit_identical_to_one_pass(precision):
    """model.point_cloud_encoder.workspace_chunk: a batch larger than the chunk goes through the encoder in near-equal
    slabs that reuse one workspace.  Same output bits, same hit counts, and the peak memory of the forward drops."""
    from mpinets_amd.model import MotionPolicyNetwork
    from mpinets_amd.scenes import make_problem_batch

    torch.manual_seed(8)
    mdl = MotionPolicyNetwork().to(dev()).eval().set_precision(precision)
    B = 3 * 1100 + 7  # (slabs above 1024 rows: the dense layers keep the launch shape of the whole batch)
    prob = make_problem_batch(B, seed=43, device=dev(), kinds=("tabletop", "cubby", "dresser"), M1=40, scene_pool=64,
                              device_clouds=True)
    enc = mdl.point_cloud_encoder
    with torch.no_grad():
        enc.workspace_chunk = None
        mdl(prob["xyz"][:8], prob["q_norm"][:8])  # (weight packs built)
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        one = mdl(prob["xyz"], prob["q_norm"]).clone()
        c_one = [c.clone() for c in enc.last_counts]
        torch.cuda.synchronize()
        peak_one = torch.cuda.max_memory_allocated() - base
        enc.workspace_chunk = 1200  # -> three slabs of 1102 / 1103 environments
        torch.cuda.reset_peak_memory_stats()
        slabs = mdl(prob["xyz"], prob["q_norm"]).clone()
        torch.cuda.synchronize()
        peak_slabs = torch.cuda.max_memory_allocated() - base
    assert torch.equal(one, slabs)
    assert all(torch.equal(a, b) for a, b in zip(c_one, enc.last_counts))
    print(f"peak forward memory: one pass {peak_one / 2**20:.0f} MiB, slabs {peak_slabs / 2**20:.0f} MiB")
    assert peak_slabs < 0.45 * peak_one
