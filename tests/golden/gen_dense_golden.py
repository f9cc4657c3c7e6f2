"""Generates tests/golden/dense_golden.npz by IMPORTING AND RUNNING the reference's mpinets/model.py on DENSE clouds.

``model_golden.npz`` (gen_model_golden.py) uses realistic scenes: their set-abstraction neighbourhoods never fill (at most
56 of 128 hits in the first module, 0 of 1536 queries saturated), so the ball query's truncation rule -- more hits than
slots: the FIRST ``nsample`` by index -- and the grouped-MLP kernels' behaviour on FULL rows (no padding to elide) met the
reference's composition only through the oracle.  Here every environment is built to saturate both modules:

* the 4096 scene points lie on cuboids of at most 10 cm (a ball of 5 cm holds hundreds of them);
* the robot is folded onto itself (joints near their limits: most of its 2048 points within 30 cm of each other), so the
  second module's 30 cm balls hold more than 128 of the 512 sampled points;
* one ordinary tabletop environment rides along, so saturated and sparse rows share a batch.

Same stubs as gen_model_golden.py (imported from it): the indices and the FK come from this repo's oracle (pointnet2_ops /
robofin are absent -- parity unpinned); grouping, the three shared MLPs, max-pools, fc layer, heads and
the rollout loop are the reference's own code (model.py:75-91, 128-183, 366-383, 409-426).

    python tests/golden/gen_dense_golden.py
"""
import os
import sys

sys.dont_write_bytecode = True

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_model_golden as gm  # noqa: E402  (path set-up, stubs, seeded weights, make_slabs)

from mpinets_amd import franka_tables as ft  # noqa: E402
from mpinets_amd import scenes  # noqa: E402
from oracle import oracle  # noqa: E402
import seeded_weights  # noqa: E402

NR, NS, NT = gm.NR, gm.NS, gm.NT
M1, M2 = 8, 4


def folded_configurations():
    """Two folded arm poses: from a small grid of near-limit joint values, the ones whose 2048-point cloud has the smallest
    mean distance to its centroid (deterministic; nothing random)."""
    lim = ft.JOINT_LIMITS_REAL.astype(np.float64)
    pts, link = ft.link_point_table(4096, True)
    sub = np.arange(0, 4096, 2, dtype=np.int32)
    cands = []
    for j2 in (-1.45, -0.9, 0.9, 1.45):
        for j4 in (-2.7, -2.4):
            for j6 in (0.9, 2.4, 4.1):
                for j1 in (0.0, 1.2):
                    cands.append([j1, j2, 0.3, j4, -0.4, j6, 0.8])
    q = np.clip(np.array(cands, np.float64), lim[:, 0] + 1e-3, lim[:, 1] - 1e-3).astype(np.float32)
    cloud = oracle.transform_table(oracle.franka_fk(q), pts, link, sub)
    spread = np.linalg.norm(cloud - cloud.mean(1, keepdims=True), axis=2).mean(1)
    order = np.argsort(spread, kind="stable")
    return q[order[[0, 3]]], spread[order[[0, 3]]]


def dense_scene(cuboids):
    """Scene dict of one environment: the given small cuboids (centre, dims, yaw), every other row zero-volume like the
    data loader pads them (data_loader.py:202)."""
    s = {"cuboid_centers": np.zeros((1, M1, 3), np.float32), "cuboid_dims": np.zeros((1, M1, 3), np.float32),
         "cuboid_quats": np.tile(np.array([1, 0, 0, 0], np.float32), (1, M1, 1)),
         "cylinder_centers": np.zeros((1, M2, 3), np.float32), "cylinder_radii": np.zeros((1, M2, 1), np.float32),
         "cylinder_heights": np.zeros((1, M2, 1), np.float32),
         "cylinder_quats": np.tile(np.array([1, 0, 0, 0], np.float32), (1, M2, 1))}
    for i, (c, d, yaw) in enumerate(cuboids):
        s["cuboid_centers"][0, i], s["cuboid_dims"][0, i] = c, d
        s["cuboid_quats"][0, i] = [np.cos(yaw / 2), 0, 0, np.sin(yaw / 2)]
    return s


def hit_counts(new_xyz, xyz, radius):
    """Brute-force number of points with d2 < r2 per query (float32 differences like the kernels; only used to REPORT how
    many neighbourhoods overflow -- the indices themselves come from oracle.ball_query)."""
    d = new_xyz[:, :, None, :].astype(np.float32) - xyz[:, None, :, :].astype(np.float32)
    return ((d * d).sum(-1) < np.float32(radius) ** 2).sum(-1)


def main():
    gm.install_stubs()
    import mpinets.model as ref_model

    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    mdl = ref_model.TrainingMotionPolicyNetwork(NR, 1.0, 1.0)
    shapes = {k: tuple(v.shape) for k, v in mdl.state_dict().items()}
    sd = seeded_weights.seeded_state_dict(shapes, seed=0)
    mdl.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()}, strict=True)
    mdl.eval()

    # ---- inputs: two dense environments + one ordinary tabletop
    qf, spread = folded_configurations()
    print("folded poses:", qf.round(2).tolist(), "mean distance to centroid", spread.round(3).tolist())
    dense = [dense_scene([((0.32, 0.05, 0.30), (0.08, 0.08, 0.08), 0.3)]),
             dense_scene([((0.28, -0.10, 0.45), (0.10, 0.04, 0.04), 0.9), ((0.30, -0.02, 0.40), (0.06, 0.06, 0.06), 0.0)])]
    xyz_t, q_t, qn_t, _, scn_t = gm.make_slabs(1, 23, ("tabletop",))
    lim = ft.JOINT_LIMITS_REAL.astype(np.float32)
    pts, link = ft.link_point_table(4096, True)
    rng = np.random.default_rng(31)
    xyz = np.zeros((3, NR + NS + NT, 4), np.float32)
    xyz[:, NR:NR + NS, 3] = 1
    xyz[:, NR + NS:, 3] = 2
    q = np.concatenate([qf, q_t]).astype(np.float32)
    for b in range(2):
        sub = rng.permutation(4096)[:NR].astype(np.int32)
        xyz[b, :NR, :3] = oracle.transform_table(oracle.franka_fk(q[b:b + 1]), pts, link, sub)[0]
        xyz[b, NR:NR + NS, :3] = scenes.sample_scene_clouds_host(dense[b], NS, 40 + b)[0]
        # the target gripper sits next to the obstacle: its 128 points are dense too
        qt = np.clip(q[b:b + 1] + np.float32(0.15), lim[:, 0], lim[:, 1]).astype(np.float32)
        pose = oracle.frames_to_4x4(oracle.franka_fk(qt)[:, ft.LINK_ID["right_gripper"]])[0]
        eef = ft.end_effector_point_table()[rng.permutation(512)[:NT]]
        xyz[b, NR + NS:, :3] = eef @ pose[:3, :3].T + pose[:3, 3]
    xyz[2] = xyz_t[0]
    qn = ((q - lim[:, 0]) / (lim[:, 1] - lim[:, 0]) * 2 - 1).astype(np.float32)
    scn = {k: np.concatenate([dense[0][k], dense[1][k], scn_t[k][:, :M1] if k.startswith("cuboid") else scn_t[k][:, :M2]])
           for k in gm.SCENE_KEYS}

    # ---- forward with every module's output, and the full index rows of both ball queries
    dq = mdl(torch.as_tensor(xyz), torch.as_tensor(qn))
    sa = mdl.point_cloud_encoder.SA_modules
    xyz1, xyz2 = sa[0].last["new_xyz"].numpy(), sa[1].last["new_xyz"].numpy()
    ball1 = oracle.ball_query(xyz1, np.ascontiguousarray(xyz[:, :, :3]), 0.05, 128)  # (the calls the stub made)
    ball2 = oracle.ball_query(xyz2, xyz1, 0.3, 128)
    hits1, hits2 = hit_counts(xyz1, xyz[:, :, :3], 0.05), hit_counts(xyz2, xyz1, 0.3)
    out = {
        "param_sha256": np.array(seeded_weights.digest(sd)),
        "d_xyz": xyz, "d_q": qn, "d_out": dq.numpy(),
        "d_fps1": sa[0].last["fps_idx"], "d_xyz1": xyz1, "d_feat1": sa[0].last["new_features"].numpy(),
        "d_fps2": sa[1].last["fps_idx"], "d_xyz2": xyz2, "d_feat2": sa[1].last["new_features"].numpy(),
        "d_feat3": sa[2].last["new_features"].numpy(),
        "d_encoding": mdl.point_cloud_encoder(torch.as_tensor(xyz)).numpy(),
        "d_ball1": ball1.astype(np.int16), "d_ball2": ball2.astype(np.int16),
        "d_hits1": hits1.astype(np.int16), "d_hits2": hits2.astype(np.int16),
    }
    for k in gm.SCENE_KEYS:
        out["d_" + k] = scn[k]
    sat1, sat2 = (hits1 > 128).sum(1), (hits2 > 128).sum(1)
    print("SA1 queries with more than 128 hits per environment:", sat1.tolist(), "of 512; max hits", hits1.max(1).tolist())
    print("SA2 queries with more than 128 hits per environment:", sat2.tolist(), "of 128; max hits", hits2.max(1).tolist())
    assert (sat1[:2] >= 20).all() and (sat2[:2] >= 20).all(), "the dense environments must overflow both modules"
    assert sat1[2] == 0 and sat2[2] == 0, "the tabletop environment is the sparse row of the batch"
    print("forward:", np.abs(out["d_out"]).max(), np.abs(out["d_encoding"]).max())

    # ---- 5-step rollout (model.py:128-183) from the dense slabs: the robot unfolds or stays, the scene rows stay dense
    sampler = gm._FrankaSampler("cpu")
    np.random.seed(13)
    del gm.SUBSETS[:]
    slab = torch.as_tensor(xyz.copy())
    traj = mdl.rollout({"xyz": slab, "configuration": torch.as_tensor(qn.copy())}, 5, lambda qq: sampler.sample(qq, NR))
    out.update(d_traj=torch.stack(traj).numpy(), d_robot=slab[:, :NR, :3].numpy().copy(), d_subsets=np.stack(gm.SUBSETS))
    print("rollout: max step", np.abs(np.diff(out["d_traj"], axis=0)).max())

    path = os.path.join(HERE, "dense_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote dense_golden.npz:", os.path.getsize(path) // 1024, "KB")


if __name__ == "__main__":
    main()
