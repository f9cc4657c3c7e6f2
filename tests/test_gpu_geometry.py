"""HIP geometry (C-ABI via the Python mirror) vs the reference's golden vectors and the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 1e-5  # north-star tolerance for fp32 SDFs (BASELINE.json)
SUITES = ["tabletop_yaw", "cubby_yaw_padded", "full_rotation_quirk", "unnormalised_quats", "all_masked", "single_prim"]


def dev():
    return torch.device("cuda:0")


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


@pytest.mark.parametrize("suite", SUITES)
def test_golden_vectors(golden, suite):
    from mpinets_amd.geometry import TorchCuboids, TorchCylinders, TorchSpheres

    g = lambda k: golden[f"{suite}/{k}"]
    cub = TorchCuboids(T(g("cub_centers")), T(g("cub_dims")), T(g("cub_quats")))
    cyl = TorchCylinders(T(g("cyl_centers")), T(g("cyl_radii")), T(g("cyl_heights")), T(g("cyl_quats")))
    sph = TorchSpheres(T(g("sph_centers")), T(g("sph_radii")))
    np.testing.assert_allclose(cub.inv_frames.cpu().numpy(), g("out/cub_inv_frames"), rtol=0, atol=TOL)
    np.testing.assert_allclose(cyl.inv_frames.cpu().numpy(), g("out/cyl_inv_frames"), rtol=0, atol=TOL)
    np.testing.assert_array_equal(cub.mask.cpu().numpy(), g("out/cub_mask"))
    np.testing.assert_array_equal(cyl.mask.cpu().numpy(), g("out/cyl_mask"))
    np.testing.assert_array_equal(sph.mask.cpu().numpy(), g("out/sph_mask"))
    for obj, tag in ((cub, "cub"), (cyl, "cyl"), (sph, "sph")):
        np.testing.assert_allclose(obj.sdf(T(g("points"))).cpu().numpy(), g(f"out/{tag}_sdf"), rtol=0, atol=TOL)
        np.testing.assert_allclose(obj.sdf_sequence(T(g("seq"))).cpu().numpy(), g(f"out/{tag}_sdf_seq"), rtol=0, atol=TOL)


@pytest.mark.parametrize("suite", ["tabletop_yaw", "cubby_yaw_padded"])
def test_sphere_sample_surface_and_area_golden(golden, suite):
    """TorchSpheres.sample_surface / surface_area (geometry.py:60-85) vs vectors made by running the reference under
    torch.manual_seed(11): the same draw from torch's global CPU generator, the same normalise-and-scale arithmetic
    (zero-radius spheres included: their samples collapse onto the centre)."""
    from mpinets_amd.geometry import TorchSpheres

    g = lambda k: golden[f"{suite}/{k}"]
    sph = TorchSpheres(T(g("sph_centers")), T(g("sph_radii")))
    torch.manual_seed(11)
    pts = sph.sample_surface(7)
    want = g("out/sph_surface_points_seed11_n7")
    assert tuple(pts.shape) == want.shape and pts.device.type == "cuda"
    np.testing.assert_allclose(pts.cpu().numpy(), want, rtol=0, atol=1e-6)
    np.testing.assert_allclose(sph.surface_area().cpu().numpy(), g("out/sph_surface_area"), rtol=1e-6, atol=0)
    # every sample lies on its sphere: |sdf of the sphere alone| ~ 0
    d = torch.linalg.norm(pts - sph.centers[:, :, None, :], dim=-1) - sph.radii
    assert float(d.abs().max()) < 1e-6


def test_matches_oracle_bitwise_on_random_scenes(oracle):
    """Same fp32 operation order as the oracle -> expect (and report) ulp-level agreement."""
    from mpinets_amd.geometry import TorchCuboids, TorchCylinders
    from mpinets_amd.scenes import make_scenes

    scn = make_scenes(48, 3, ("tabletop", "cubby", "dresser"), 40, 16)
    rng = np.random.default_rng(0)
    pts = rng.uniform(-1.5, 1.5, (48, 2048, 3)).astype(np.float32)
    cub = TorchCuboids(T(scn["cuboid_centers"]), T(scn["cuboid_dims"]), T(scn["cuboid_quats"]))
    cyl = TorchCylinders(T(scn["cylinder_centers"]), T(scn["cylinder_radii"]), T(scn["cylinder_heights"]),
                         T(scn["cylinder_quats"]))
    a = cub.sdf(T(pts)).cpu().numpy()
    b = cyl.sdf(T(pts)).cpu().numpy()
    oa = oracle.cuboid_sdf(scn["cuboid_centers"], scn["cuboid_dims"], scn["cuboid_quats"], pts)
    ob = oracle.cylinder_sdf(scn["cylinder_centers"], scn["cylinder_radii"], scn["cylinder_heights"],
                             scn["cylinder_quats"], pts)
    np.testing.assert_allclose(a, oa, rtol=0, atol=1e-6)
    np.testing.assert_allclose(b, ob, rtol=0, atol=1e-6)
    print("bitwise-equal fraction: cuboid %.4f cylinder %.4f" % ((a == oa).mean(), (b == ob).mean()))


def test_sequence_equals_slices_at_full_size():
    """Size-independent property at BASELINE scale (C4 shape: B=1024/GPU, T=50, S=56)."""
    from mpinets_amd.geometry import TorchCuboids
    from mpinets_amd.scenes import make_scenes

    B, Tn, S = 1024, 50, 56
    scn = make_scenes(64, 5, ("tabletop",), 16, 16)
    rep = lambda a: np.tile(a, (B // 64,) + (1,) * (a.ndim - 1))
    cub = TorchCuboids(T(rep(scn["cuboid_centers"])), T(rep(scn["cuboid_dims"])), T(rep(scn["cuboid_quats"])))
    seq = torch.rand((B, Tn, S, 3), device=dev()) * 2 - 1
    whole = cub.sdf_sequence(seq)
    assert whole.shape == (B, Tn, S)
    for t in (0, 17, 49):
        assert torch.equal(whole[:, t], cub.sdf(seq[:, t].contiguous()))


def test_empty_and_cpu_inputs_rejected():
    from mpinets_amd import _lib
    from mpinets_amd.geometry import TorchCuboids

    with pytest.raises(_lib.MpxError):
        TorchCuboids(torch.zeros(1, 1, 3), torch.ones(1, 1, 3), torch.tensor([[[1.0, 0, 0, 0]]]))
    cub = TorchCuboids(torch.zeros(2, 1, 3, device=dev()), torch.zeros(2, 1, 3, device=dev()),
                       torch.tensor([[[1.0, 0, 0, 0]]], device=dev()).repeat(2, 1, 1))
    out = cub.sdf(torch.rand(2, 7, 3, device=dev()))
    assert torch.isinf(out).all() and out.shape == (2, 7)
