"""Scene-cloud sampler: the oracle's restatement follows the distribution of the reference's
construct_mixed_point_cloud (CPU); the HIP kernel equals the oracle bit-for-bit on ids (GPU)."""
import random

import numpy as np
import pytest
import torch


def _scene(B=1, seed=3, kinds=("tabletop",), M1=16, M2=16):
    from mpinets_amd.scenes import make_scenes

    return make_scenes(B, seed, kinds, M1, M2)


def _primitives(scn, b):
    from mpinets_amd.primitives import Cuboid, Cylinder

    cubs = [Cuboid(c, d, q) for c, d, q in zip(scn["cuboid_centers"][b], scn["cuboid_dims"][b], scn["cuboid_quats"][b])]
    cubs = [c for c in cubs if not c.is_zero_volume()]
    cyls = [Cylinder(c, r[0], h[0], q) for c, r, h, q in zip(scn["cylinder_centers"][b], scn["cylinder_radii"][b],
                                                             scn["cylinder_heights"][b], scn["cylinder_quats"][b])]
    cyls = [c for c in cyls if not c.is_zero_volume()]
    return cubs + cyls  # data_loader.py:258 order


def test_philox_known_answers(oracle):
    """Random123 known-answer vectors for Philox4x32-10."""
    assert [hex(x) for x in oracle.philox4x32([0, 0, 0, 0], [0, 0])] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    assert [hex(x) for x in oracle.philox4x32([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2)] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]


def test_oracle_points_lie_on_their_obstacle(oracle):
    scn = _scene(3, 5, ("tabletop", "cubby", "dresser"), 40, 16)
    pts, assign, labels, nobs = oracle.scene_cloud(scn, 4096, 11)
    for b in range(3):
        prims = _primitives(scn, b)
        assert nobs[b] == len(prims)
        live = [m for m in range(56) if labels[b, m] != 0]
        assert sorted(labels[b, live]) == list(range(1, len(prims) + 1))  # labels are a permutation of 1..K
        for k, m in enumerate(live):
            p = pts[b][assign[b] == m]
            assert len(p) > 0 and np.abs(prims[k].sdf(p)).max() < 2e-6


def test_distribution_matches_reference_function(oracle):
    """Per-obstacle counts: mean over many draws of the oracle's sampler vs the reference algorithm
    (mpinets_amd.geometry.construct_mixed_point_cloud, pinned to geometry.py:571-608 by the golden test)."""
    from mpinets_amd.geometry import construct_mixed_point_cloud

    scn1 = _scene(1, 8)
    prims = _primitives(scn1, 0)
    K, N, R = len(prims), 4096, 192
    scn = {k: np.repeat(v, R, axis=0) for k, v in scn1.items()}
    _, assign, labels, _ = oracle.scene_cloud(scn, N, 2024)
    live = [m for m in range(32) if labels[0, m] != 0]
    mine = np.stack([(assign == m).sum(1) for m in live], 1).astype(np.float64)  # [R,K]
    random.seed(1)
    np.random.seed(1)
    ref = []
    for _ in range(48):
        pc = construct_mixed_point_cloud(prims, N)
        # labels are shuffled per call: a label's obstacle is the one ALL of its points lie on
        cnt = np.zeros(K)
        for lab in np.unique(pc[:, 3]):
            pts = pc[pc[:, 3] == lab, :3]
            k = int(np.argmin([np.abs(p.sdf(pts)).max() for p in prims]))
            cnt[k] += len(pts)
        ref.append(cnt)
    ref = np.asarray(ref, dtype=np.float64)
    areas = np.array([p.surface_area for p in prims])
    pool = np.floor(areas / areas.sum() * N) + 500
    expect = N * pool / pool.sum()  # hypergeometric mean of geometry.py:598-608
    np.testing.assert_allclose(mine.mean(0), expect, rtol=0.03)
    np.testing.assert_allclose(ref.mean(0), expect, rtol=0.04)
    # spread: hypergeometric variance, not multinomial
    T = pool.sum()
    var = N * (pool / T) * (1 - pool / T) * (T - N) / (T - 1)
    np.testing.assert_allclose(mine.var(0), var, rtol=0.35)
    # labels: uniform over permutations -> every label value appears for obstacle 0 across environments
    assert len(set(labels[:, live[0]])) >= min(K, 5)


def test_oracle_empty_and_determinism(oracle):
    scn = _scene(2, 1)
    scn["cuboid_dims"][1] = 0
    scn["cylinder_radii"][1] = 0
    a = oracle.scene_cloud(scn, 512, 7)
    b = oracle.scene_cloud(scn, 512, 7)
    c = oracle.scene_cloud(scn, 512, 8)
    assert all(np.array_equal(x, y) for x, y in zip(a, b)) and not np.array_equal(a[1][0], c[1][0])
    assert a[3][1] == 0 and (a[1][1] == 0xFFFF).all() and (a[0][1] == 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("kinds,M1,M2,N", [(("tabletop",), 16, 16, 4096), (("tabletop", "cubby", "dresser"), 40, 16, 4096),
                                          (("cubby",), 8, 1, 1000)])
def test_device_sampler_equals_oracle(oracle, kinds, M1, M2, N):
    from mpinets_amd.scenes import sample_scene_clouds

    dev = torch.device("cuda:0")
    B = 70
    scn = _scene(B, 21, kinds, M1, M2)
    scn["cuboid_dims"][3] = 0  # one environment with no cuboids
    if M2 > 1:
        scn["cylinder_radii"][3] = 0  # ... and no cylinders: empty scene
    prims = {k: torch.from_numpy(v).to(dev) for k, v in scn.items()}
    out, assign, labels, nobs = sample_scene_clouds(prims, N, seed=1234567, return_aux=True)
    opts, oassign, olabels, onobs = oracle.scene_cloud(scn, N, 1234567)
    np.testing.assert_array_equal(assign.cpu().numpy().view(np.uint16), oassign)
    np.testing.assert_array_equal(labels.cpu().numpy(), olabels)
    np.testing.assert_array_equal(nobs.cpu().numpy(), onobs)
    np.testing.assert_allclose(out.cpu().numpy(), opts, rtol=0, atol=1e-6)
    # in place into a slab view with labels, nothing else touched
    slab = torch.full((B, 2048 + N + 128, 4), -3.0, device=dev)
    sample_scene_clouds(prims, N, seed=1234567, out=slab[:, 2048:2048 + N], write_label=True)
    np.testing.assert_array_equal(slab[:, 2048:2048 + N, :3].cpu().numpy(), out.cpu().numpy())
    lab = slab[:, 2048:2048 + N, 3].cpu().numpy()
    want = np.take_along_axis(olabels.astype(np.float32), np.minimum(oassign, M1 + M2 - 1).astype(np.int64), 1)
    want[oassign == 0xFFFF] = 0
    np.testing.assert_array_equal(lab, want)
    assert (slab[:, :2048] == -3).all() and (slab[:, 2048 + N:] == -3).all()


@pytest.mark.gpu
def test_device_clouds_feed_the_policy(oracle):
    """make_problem_batch(device_clouds=True): scene rows come from the kernel and lie on the primitives."""
    from mpinets_amd.geometry import TorchCuboids, TorchCylinders
    from mpinets_amd.scenes import make_problem_batch

    prob = make_problem_batch(6, seed=2, device="cuda:0", device_clouds=True)
    pts = prob["xyz"][:, 2048:6144, :3].cpu().numpy()
    scn = {k: v.cpu().numpy() for k, v in prob.items() if k.startswith(("cuboid_", "cylinder_"))}
    for b in range(6):
        # objects may interpenetrate, so test "on SOME obstacle's surface", not the scene SDF
        d = np.min([np.abs(p.sdf(pts[b])) for p in _primitives(scn, b)], axis=0)
        assert d.max() < 2e-6
    assert (prob["xyz"][:, 2048:6144, 3] == 1).all()
    # and the clouds differ between environments that share primitives (scene_pool tiling)
    prob2 = make_problem_batch(4, seed=2, device="cuda:0", device_clouds=True, scene_pool=1)
    assert not torch.equal(prob2["xyz"][0, 2048:6144], prob2["xyz"][1, 2048:6144])


def test_oracle_subset_draw_is_a_uniform_draw_without_replacement(oracle):
    """orc_draw_subset (the per-call robot-point subset, robofin FrankaSampler.sample's np.random.choice(P, n, replace=False),
    model.py:170-181): distinct rows in range, a function of (seed, draw) only, every row about equally likely, and
    n_out = total is a permutation."""
    a = oracle.draw_subset(4096, 2048, 5, 0)
    assert a.dtype == np.int32 and len(np.unique(a)) == 2048 and a.min() >= 0 and a.max() < 4096
    np.testing.assert_array_equal(a, oracle.draw_subset(4096, 2048, 5, 0))
    assert not np.array_equal(a, oracle.draw_subset(4096, 2048, 5, 1))
    assert not np.array_equal(a, oracle.draw_subset(4096, 2048, 6, 0))
    np.testing.assert_array_equal(np.sort(oracle.draw_subset(77, 77, 1, 3)), np.arange(77))
    hits = np.zeros(64)
    for d in range(400):  # 16 of 64 rows, 400 draws: each row is chosen 100 times on average
        hits[oracle.draw_subset(64, 16, 9, d)] += 1
    assert hits.min() > 60 and hits.max() < 140, hits
    first = np.bincount([oracle.draw_subset(64, 16, 9, d)[0] for d in range(400)], minlength=64)
    assert first.max() < 25  # (uniform ORDER too: no row is favoured as the first pick)
