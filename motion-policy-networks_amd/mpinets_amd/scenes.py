"""Synthetic planning problems for tests and benchmarks.

The reference's scene generators (``mpinets/data_pipeline/environments/*.py``) need PyBullet and
IK and are out of scope; this module restates their *distributions* with a seeded NumPy RNG:

* tabletop -- ``tabletop_environment.py:215-324,406-441``: table height 0 w.p. 0.35 else U(0,0.4);
  front table x in [U(.275,.375), U(1.275,1.375)], width U(1.5,1.65); optional side table p=0.5;
  K ~ randint(3,15) objects, 30 % cylinders r in [.05,.15], h in [.05,.35], else cuboids
  xy in [.05,.15], z in [.05,.35], yaw U(0, pi/2);
* cubby -- ``cubby_environment.py:62-74``: a 2x2 shelf made of 7 thin cuboids (thickness
  U(.01,.03)), yaw +-pi/18;
* dresser-like -- ``dresser_environment.py:198-222``: up to 40 yaw-rotated cuboids.

Primitive sets are zero-padded to fixed M1 / M2 exactly like the dataset rows
(zero dims, identity quaternion, ``mpinets/data_loader.py:202,210-215``).
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from . import franka_tables as ft

NUM_ROBOT_POINTS = 2048
NUM_OBSTACLE_POINTS = 4096
NUM_TARGET_POINTS = 128


def _yaw_quat(yaw):
    q = np.zeros(np.shape(yaw) + (4,))
    q[..., 0] = np.cos(np.asarray(yaw) / 2)
    q[..., 3] = np.sin(np.asarray(yaw) / 2)
    return q


def _tabletop(rng, M1, M2):
    cc, cd, cq = np.zeros((M1, 3)), np.zeros((M1, 3)), np.tile([1.0, 0, 0, 0], (M1, 1))
    yc, yr, yh, yq = np.zeros((M2, 3)), np.zeros((M2, 1)), np.zeros((M2, 1)), np.tile([1.0, 0, 0, 0], (M2, 1))
    h = 0.0 if rng.random() < 0.35 else rng.uniform(0.0, 0.4)
    x0, x1 = rng.uniform(0.275, 0.375), rng.uniform(1.275, 1.375)
    w = rng.uniform(1.5, 1.65)
    thick = 0.05
    n = 0
    cc[n], cd[n] = [(x0 + x1) / 2, 0.0, h - thick / 2], [x1 - x0, w, thick]
    n += 1
    if rng.random() < 0.5:  # side table
        side = rng.choice([-1.0, 1.0])
        cc[n], cd[n] = [0.0, side * (w / 4 + 0.45), h - thick / 2], [0.9, w / 2, thick]
        n += 1
    cc[n], cd[n] = [-0.35, 0.0, -0.025], [0.6, 0.6, 0.05]  # mount table under the robot
    n += 1
    K = int(rng.integers(3, 15))
    m = 0
    for _ in range(K):
        px, py = rng.uniform(x0 + 0.1, x1 - 0.1), rng.uniform(-w / 2 + 0.1, w / 2 - 0.1)
        if rng.random() < 0.3 and m < M2:
            r, hh = rng.uniform(0.05, 0.15), rng.uniform(0.05, 0.35)
            yc[m], yr[m], yh[m] = [px, py, h + hh / 2], r, hh
            m += 1
        elif n < M1:
            d = np.array([rng.uniform(0.05, 0.15), rng.uniform(0.05, 0.15), rng.uniform(0.05, 0.35)])
            cc[n], cd[n], cq[n] = [px, py, h + d[2] / 2], d, _yaw_quat(rng.uniform(0, np.pi / 2))
            n += 1
    return cc, cd, cq, yc, yr, yh, yq


def _cubby(rng, M1, M2):
    cc, cd, cq = np.zeros((M1, 3)), np.zeros((M1, 3)), np.tile([1.0, 0, 0, 0], (M1, 1))
    yc, yr, yh, yq = np.zeros((M2, 3)), np.zeros((M2, 1)), np.zeros((M2, 1)), np.tile([1.0, 0, 0, 0], (M2, 1))
    t = rng.uniform(0.01, 0.03)
    W, H, D = rng.uniform(0.7, 1.1), rng.uniform(0.5, 0.9), rng.uniform(0.2, 0.35)
    cx, cz = rng.uniform(0.55, 0.8), rng.uniform(0.1, 0.3)
    yaw = rng.uniform(-np.pi / 18, np.pi / 18)
    c, s = np.cos(yaw), np.sin(yaw)
    R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])
    parts = [  # local centre, dims (x = depth, y = width, z = height)
        ([0, 0, 0], [D, W, t]), ([0, 0, H], [D, W, t]), ([0, 0, H / 2], [D, W, t]),  # bottom, top, mid shelf
        ([0, -W / 2, H / 2], [D, t, H]), ([0, W / 2, H / 2], [D, t, H]), ([0, 0, H / 2], [D, t, H]),  # sides, divider
        ([D / 2, 0, H / 2], [t, W, H]),  # back
    ]
    for i, (lc, dims) in enumerate(parts[:M1]):
        cc[i] = R @ np.array(lc) + np.array([cx, 0, cz])
        cd[i] = dims
        cq[i] = _yaw_quat(yaw)
    return cc, cd, cq, yc, yr, yh, yq


def _dresser(rng, M1, M2):
    cc, cd, cq = np.zeros((M1, 3)), np.zeros((M1, 3)), np.tile([1.0, 0, 0, 0], (M1, 1))
    yc, yr, yh, yq = np.zeros((M2, 3)), np.zeros((M2, 1)), np.zeros((M2, 1)), np.tile([1.0, 0, 0, 0], (M2, 1))
    W, D, H = 1.0 + rng.uniform(-0.2, 0.2), 0.3 + rng.uniform(-0.1, 0.1), 0.7 + rng.uniform(-0.15, 0.15)
    cx = 0.65 + rng.uniform(-0.1, 0.1)
    yaw = np.pi / 2 + rng.uniform(-np.pi / 3, np.pi / 3)
    c, s = np.cos(yaw), np.sin(yaw)
    R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])
    n = min(M1, int(rng.integers(12, 41)))
    rows = int(np.ceil(np.sqrt(n)))
    for i in range(n):
        r_, c_ = divmod(i, rows)
        w_, h_ = W / rows, H / rows
        lc = np.array([(c_ + 0.5) * w_ - W / 2, rng.uniform(-0.02, 0.02), (r_ + 0.5) * h_])
        cc[i] = R @ lc + np.array([cx, 0, 0])
        cd[i] = [w_ * 0.95, D, h_ * 0.2]
        cq[i] = _yaw_quat(yaw)
    return cc, cd, cq, yc, yr, yh, yq


_GENERATORS = {"tabletop": _tabletop, "cubby": _cubby, "dresser": _dresser}


def make_scenes(B: int, seed: int = 0, kinds=("tabletop",), M1: int = 16, M2: int = 16) -> Dict[str, np.ndarray]:
    """-> float32 arrays cuboid_{centers,dims,quats} [B,M1,*], cylinder_{centers,radii,heights,quats} [B,M2,*]."""
    rng = np.random.default_rng(seed)
    out = [[] for _ in range(7)]
    for b in range(B):
        parts = _GENERATORS[kinds[b % len(kinds)]](rng, M1, M2)
        for o, p in zip(out, parts):
            o.append(p)
    names = ["cuboid_centers", "cuboid_dims", "cuboid_quats", "cylinder_centers", "cylinder_radii",
             "cylinder_heights", "cylinder_quats"]
    return {k: np.asarray(v, dtype=np.float32) for k, v in zip(names, out)}


def sample_scene_clouds_host(scn: Dict[str, np.ndarray], num_points: int, seed: int = 0) -> np.ndarray:
    """Vectorised area-proportional surface sampling of every environment's primitives (NumPy).

    Distributionally equivalent to ``construct_mixed_point_cloud`` (geometry.py:571-608): each
    point picks an unmasked primitive with probability proportional to its surface area and is
    uniform on that surface.  -> float32 [B, num_points, 3].  Host-side set-up code (the batched
    device sampler is part of the S3 row).  All uniforms come from ONE array drawn scene by scene,
    so the cloud of scene b does not depend on how many scenes follow it (a shard that generates a prefix of the
    scenes gets the same clouds as a process that generates them all).
    """
    rng = np.random.default_rng(seed)
    U = rng.random((scn["cuboid_dims"].shape[0], num_points, 10))
    cd, yr, yh = scn["cuboid_dims"].astype(np.float64), scn["cylinder_radii"][..., 0].astype(np.float64), \
        scn["cylinder_heights"][..., 0].astype(np.float64)
    B, M1 = cd.shape[:2]
    M2 = yr.shape[1]
    cub_area = 2 * (cd[..., 0] * cd[..., 1] + cd[..., 0] * cd[..., 2] + cd[..., 1] * cd[..., 2])
    cub_area[(np.abs(cd) <= 1e-8).any(-1)] = 0
    cyl_area = 2 * np.pi * yr * yh + 2 * np.pi * yr**2
    cyl_area[(np.abs(yr) <= 1e-8) | (np.abs(yh) <= 1e-8)] = 0
    area = np.concatenate([cub_area, cyl_area], axis=1)
    cdf = np.cumsum(area, axis=1)
    u = U[..., 0] * cdf[:, -1:]
    prim = np.minimum((u[:, :, None] >= cdf[:, None, :]).sum(-1), M1 + M2 - 1)  # [B,P]
    bi = np.arange(B)[:, None]
    is_cyl = prim >= M1
    ci = np.minimum(prim, M1 - 1)
    yi = np.clip(prim - M1, 0, M2 - 1)
    # cuboid samples
    d = cd[bi, ci]
    face_area = np.stack([d[..., 1] * d[..., 2], d[..., 0] * d[..., 2], d[..., 0] * d[..., 1]], -1)
    fc = np.cumsum(face_area, -1)
    uf = U[..., 1] * fc[..., -1]
    axis = np.minimum((uf[..., None] >= fc).sum(-1), 2)
    p = (U[..., 2:5] - 0.5) * d
    sign = np.where(U[..., 5] < 0.5, -0.5, 0.5)
    np.put_along_axis(p, axis[..., None], (sign * np.take_along_axis(d, axis[..., None], -1)[..., 0])[..., None], -1)
    # cylinder samples
    r, h = yr[bi, yi], yh[bi, yi]
    side, cap = 2 * np.pi * r * h, np.pi * r**2
    uc = U[..., 6] * (side + 2 * cap + 1e-30)
    th = U[..., 7] * 2 * np.pi
    rho = np.where(uc < side, r, r * np.sqrt(U[..., 8]))
    z = np.where(uc < side, (U[..., 9] - 0.5) * h, np.where(uc < side + cap, -0.5 * h, 0.5 * h))
    pc = np.stack([rho * np.cos(th), rho * np.sin(th), z], -1)
    local = np.where(is_cyl[..., None], pc, p)
    quat = np.where(is_cyl[..., None], scn["cylinder_quats"][bi, yi], scn["cuboid_quats"][bi, ci]).astype(np.float64)
    ctr = np.where(is_cyl[..., None], scn["cylinder_centers"][bi, yi], scn["cuboid_centers"][bi, ci]).astype(np.float64)
    quat /= np.linalg.norm(quat, axis=-1, keepdims=True)
    w, x, y, zq = quat[..., 0], quat[..., 1], quat[..., 2], quat[..., 3]
    R = np.stack([
        np.stack([1 - 2 * (y * y + zq * zq), 2 * (x * y - w * zq), 2 * (x * zq + w * y)], -1),
        np.stack([2 * (x * y + w * zq), 1 - 2 * (x * x + zq * zq), 2 * (y * zq - w * x)], -1),
        np.stack([2 * (x * zq - w * y), 2 * (y * zq + w * x), 1 - 2 * (x * x + y * y)], -1)], -2)
    world = np.einsum("bpij,bpj->bpi", R, local) + ctr
    return world.astype(np.float32)


def sample_scene_clouds(prims: Dict[str, torch.Tensor], num_points: int, seed: int, out: Optional[torch.Tensor] = None,
                        write_label: bool = False, return_aux: bool = False, scratch: Optional[tuple] = None,
                        env_offset: int = 0):
    """Device-side batched ``construct_mixed_point_cloud`` (geometry.py:571-608; csrc/scene.hip).

    ``prims``: cuboid_{centers,dims,quats} [B,M1,*], cylinder_{centers,radii,heights,quats} [B,M2,*] on the
    GPU (zero-volume rows are skipped like data_loader.py:248,256).  Writes ``out[:, :num_points, :3]``
    (``out`` may be a slab view ``xyz[:, 2048:6144]``; default: a fresh [B,N,3] tensor), plus the
    shuffled obstacle label in column 3 when ``write_label``.  Deterministic in (seed, GLOBAL environment id =
    ``env_offset`` + row): rank r of a sharded batch passes the id of its first environment and draws exactly what a
    single process would draw for those environments.
    """
    from . import _lib

    cc, cd, cq = (_lib.f32c(prims[k]) for k in ("cuboid_centers", "cuboid_dims", "cuboid_quats"))
    yc, yr, yh, yq = (_lib.f32c(prims[k]) for k in ("cylinder_centers", "cylinder_radii", "cylinder_heights",
                                                    "cylinder_quats"))
    _lib.require_cuda(cc, yc)
    B, M1 = cd.shape[:2]
    M2 = yr.shape[1]
    dev = cc.device
    if out is None:
        out = torch.empty((B, num_points, 4 if write_label else 3), dtype=torch.float32, device=dev)
    assert out.ndim == 3 and out.size(0) == B and out.size(1) >= num_points and out.stride(2) == 1
    if scratch is None:  # (assign, labels, nobs): callers that re-render every step keep them
        scratch = (torch.empty((B, num_points), dtype=torch.int16, device=dev),
                   torch.zeros((B, M1 + M2), dtype=torch.uint8, device=dev),
                   torch.zeros(B, dtype=torch.int32, device=dev))
    assign, labels, nobs = scratch
    for b0 in range(0, B, 65535):
        nb = min(65535, B - b0)
        sl = slice(b0, b0 + nb)
        _lib.call("mpx_scene_cloud", _lib.ptr(cc[sl]), _lib.ptr(cd[sl]), _lib.ptr(cq[sl]), M1, _lib.ptr(yc[sl]),
                  _lib.ptr(yr[sl]), _lib.ptr(yh[sl]), _lib.ptr(yq[sl]), M2, nb, num_points, int(seed) & (2 ** 64 - 1),
                  int(env_offset) + b0,
                  _lib.ptr(assign[sl]), _lib.ptr(labels[sl]), _lib.ptr(nobs[sl]), _lib.ptr(out[sl]), out.stride(0),
                  out.stride(1), int(write_label))
    if return_aux:
        return out, assign, labels, nobs
    return out


def random_configurations(B: int, seed: int = 0) -> np.ndarray:
    """q ~ U(joint limits), float32 [B,7] (C2 of BASELINE.json)."""
    rng = np.random.default_rng(seed + 1000003)
    lim = ft.JOINT_LIMITS_REAL
    return (lim[:, 0] + rng.random((B, 7)) * (lim[:, 1] - lim[:, 0])).astype(np.float32)


def linear_trajectories(B: int, T: int, seed: int = 0) -> np.ndarray:
    """[B,T,7]: straight joint-space lines between two random configurations (C4)."""
    a, b = random_configurations(B, seed), random_configurations(B, seed + 1)
    s = np.linspace(0.0, 1.0, T, dtype=np.float32)[None, :, None]
    return (a[:, None] * (1 - s) + b[:, None] * s).astype(np.float32)


def make_problem_batch(B: int, seed: int = 0, device="cuda:0", kinds=("tabletop",), M1: int = 16, M2: int = 16,
                       scene_pool: Optional[int] = None, device_clouds: bool = False, env_offset: int = 0,
                       total_envs: Optional[int] = None) -> Dict[str, torch.Tensor]:
    """A batch of planning problems on ``device``: primitives, start configuration, target pose and
    the ``[B, 2048+4096+128, 4]`` slab (robot | scene | target rows, label column 0/1/2 --
    ``mpinets/data_loader.py:261-278``).  ``scene_pool`` bounds the number of distinct scenes
    generated on the host (they are tiled over the batch) to keep set-up time short.

    Sharding: the batch is rows ``[env_offset, env_offset + B)`` of a GLOBAL batch of ``total_envs`` problems
    (default ``env_offset + B``) that depends on ``seed`` only -- every rank passes the same seed and its own offset
    and gets exactly the rows one process would generate (scenes, configurations, targets and, with
    ``device_clouds``, the scene clouds, whose draws are keyed by the global environment id)."""
    from .robot import FrankaSampler, franka_fk, frames_to_matrix

    dev = torch.device(device)
    total = env_offset + B if total_envs is None else int(total_envs)
    assert 0 <= env_offset and env_offset + B <= total
    nscene = total if scene_pool is None else min(total, scene_pool)
    gid = env_offset + np.arange(B)  # global environment ids of this batch's rows
    sid = gid % nscene  # environment g sits in scene g mod nscene
    scn = make_scenes(nscene if scene_pool is not None else int(sid.max()) + 1 if B else 0, seed, kinds, M1, M2)
    cloud = None if device_clouds else sample_scene_clouds_host(scn, NUM_OBSTACLE_POINTS, seed)
    out = {k: torch.from_numpy(np.ascontiguousarray(v[sid])).to(dev) for k, v in scn.items()}
    q = torch.from_numpy(random_configurations(env_offset + B, seed)[env_offset:]).to(dev)
    q_target = torch.from_numpy(random_configurations(env_offset + B, seed + 7)[env_offset:]).to(dev)
    lim = torch.as_tensor(ft.JOINT_LIMITS_REAL, dtype=torch.float32, device=dev)
    state = np.random.get_state()
    np.random.seed(seed)
    sampler = FrankaSampler(dev)
    xyz = torch.zeros((B, NUM_ROBOT_POINTS + NUM_OBSTACLE_POINTS + NUM_TARGET_POINTS, 4), dtype=torch.float32,
                      device=dev)
    xyz[:, NUM_ROBOT_POINTS:NUM_ROBOT_POINTS + NUM_OBSTACLE_POINTS, 3] = 1
    xyz[:, NUM_ROBOT_POINTS + NUM_OBSTACLE_POINTS:, 3] = 2
    subset = sampler.draw_subset(NUM_ROBOT_POINTS)  # (one column subset for the whole global batch: same seed, same draw)
    sampler.sample_into(q, xyz, subset)
    if device_clouds:  # every environment gets its own draw, even when the primitives are tiled
        sample_scene_clouds(out, NUM_OBSTACLE_POINTS, seed, out=xyz[:, NUM_ROBOT_POINTS:NUM_ROBOT_POINTS + NUM_OBSTACLE_POINTS],
                            env_offset=env_offset)
    else:
        xyz[:, NUM_ROBOT_POINTS:NUM_ROBOT_POINTS + NUM_OBSTACLE_POINTS, :3] = torch.from_numpy(cloud[sid]).to(dev)
    target_pose = frames_to_matrix(franka_fk(q_target)[:, ft.LINK_ID["right_gripper"]])
    xyz[:, NUM_ROBOT_POINTS + NUM_OBSTACLE_POINTS:, :3] = sampler.sample_end_effector(target_pose, NUM_TARGET_POINTS)
    np.random.set_state(state)
    out.update(q=q, q_norm=(q - lim[:, 0]) / (lim[:, 1] - lim[:, 0]) * 2 - 1, xyz=xyz, target_pose=target_pose,
               target_position=target_pose[:, :3, 3].contiguous(), robot_subset=subset, env_offset=int(env_offset))
    return out
