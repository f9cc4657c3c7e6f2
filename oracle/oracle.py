"""oracle/oracle.py -- TEST INFRASTRUCTURE ONLY (numpy/ctypes face of oracle/mpn_oracle.c).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module, and only as the checker.  The product package never imports it.

Two parts:

* thin ctypes wrappers over ``libmpn_oracle.so`` (geometry SDFs pinned to the reference's
  ``mpinets/geometry.py`` via ``tests/golden``; FK / FPS / ball-query: PARITY UNPINNED, see the
  header of ``mpn_oracle.c``);
* a numpy restatement of the policy network (``mpinets/model.py:41-66,355-426`` layer
  dimensions; ``pointnet2_ops`` ``PointnetSAModule`` semantics [EXT-RECALL, PARITY UNPINNED]):
  float64 accumulation, float32 storage between layers.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Dict, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmpn_oracle.so")


def build(force: bool = False) -> str:
    """Compile the C restatement with gcc (seconds).  Returns the library path."""
    src = os.path.join(_HERE, "mpn_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libmpn_oracle.so"])
    return _LIB_PATH


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def set_threads(n: int = 0) -> int:
    """OpenMP threads of the C restatement's per-environment loops (0 = leave as is).  Returns the count in effect.
    Results do not depend on it; bench.py's cpu_baseline sets it to the host's core count and reports it."""
    return int(lib().orc_set_threads(int(n)))


def _f(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _i(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


# ---------------------------------------------------------------- geometry (pinned)
def prim_frames(centers, quats) -> np.ndarray:
    """[...,3],[...,4] -> [...,12] (R row-major 3x3, then Rt).  geometry.py:177-223."""
    c, q = _f(centers), _f(quats)
    n = c.size // 3
    out = np.empty((n, 12), np.float32)
    lib().orc_prim_frames(_p(c), _p(q), ctypes.c_int(n), _p(out))
    return out.reshape(c.shape[:-1] + (12,))


def inv_frames_4x4(centers, quats) -> np.ndarray:
    f = prim_frames(centers, quats)
    out = np.zeros(f.shape[:-1] + (4, 4), np.float32)
    out[..., :3, :3] = f[..., :9].reshape(f.shape[:-1] + (3, 3))
    out[..., :3, 3] = f[..., 9:]
    out[..., 3, 3] = 1
    return out


def _flat_points(points) -> Tuple[np.ndarray, tuple]:
    p = _f(points)
    return p.reshape(p.shape[0], -1, 3), p.shape[:-1]


def cuboid_sdf(centers, dims, quats, points) -> np.ndarray:
    """TorchCuboids.sdf / sdf_sequence.  points [B,N,3] or [B,T,N,3]."""
    c, d, q = _f(centers), _f(dims), _f(quats)
    p, oshape = _flat_points(points)
    B, M = c.shape[:2]
    out = np.empty((B, p.shape[1]), np.float32)
    lib().orc_cuboid_sdf(B, M, p.shape[1], _p(c), _p(d), _p(q), _p(p), _p(out))
    return out.reshape(oshape)


def cylinder_sdf(centers, radii, heights, quats, points) -> np.ndarray:
    c, r, h, q = _f(centers), _f(radii), _f(heights), _f(quats)
    p, oshape = _flat_points(points)
    B, M = c.shape[:2]
    out = np.empty((B, p.shape[1]), np.float32)
    lib().orc_cylinder_sdf(B, M, p.shape[1], _p(c), _p(r), _p(h), _p(q), _p(p), _p(out))
    return out.reshape(oshape)


def sphere_sdf(centers, radii, points) -> np.ndarray:
    c, r = _f(centers), _f(radii)
    p, oshape = _flat_points(points)
    B, M = c.shape[:2]
    out = np.empty((B, p.shape[1]), np.float32)
    lib().orc_sphere_sdf(B, M, p.shape[1], _p(c), _p(r), _p(p), _p(out))
    return out.reshape(oshape)


def collision_flags(centres, radii, cub, cyl) -> Tuple[np.ndarray, np.ndarray]:
    """model.py:293-314.  centres [B,T,S,3]; cub=(c,d,q); cyl=(c,r,h,q) -> (flags[B], min_sdf[B,T,S])."""
    x = _f(centres)
    B, T, S = x.shape[:3]
    rad = _f(radii)
    cc, cd, cq = map(_f, cub)
    yc, yr, yh, yq = map(_f, cyl)
    flags = np.zeros(B, np.uint8)
    msdf = np.empty((B, T, S), np.float32)
    lib().orc_collision_flags(B, T, S, _p(x), _p(rad), cc.shape[1], _p(cc), _p(cd), _p(cq),
                              yc.shape[1], _p(yc), _p(yr), _p(yh), _p(yq), _p(flags), _p(msdf))
    return flags.astype(bool), msdf


# ---------------------------------------------------------------- Franka FK (unpinned)
def sincos(x) -> Tuple[np.ndarray, np.ndarray]:
    x = _f(x)
    s, c = np.empty_like(x), np.empty_like(x)
    lib().orc_sincos(_p(x), ctypes.c_int(x.size), _p(s), _p(c))
    return s, c


RIGHT_GRIPPER_FRAME = 14  # last of the 15 frames orc_franka_fk writes (mpn_oracle.c)


def franka_fk(q, finger: float = 0.025) -> np.ndarray:
    """q [B,7] -> frames [B,15,12] (R row-major, t)."""
    q = _f(q).reshape(-1, 7)
    T = np.empty((q.shape[0], 15, 12), np.float32)
    lib().orc_franka_fk(_p(q), ctypes.c_int(q.shape[0]), ctypes.c_float(finger), _p(T))
    return T


def transform_table(T, pts, link_ids, subset=None) -> np.ndarray:
    T = _f(T)
    pts, link_ids = _f(pts), _i(link_ids)
    sub = None if subset is None else _i(subset)
    n_out = len(pts) if sub is None else len(sub)
    out = np.empty((T.shape[0], n_out, 3), np.float32)
    lib().orc_transform_table(_p(T), T.shape[0], T.shape[1], _p(pts), _p(link_ids), _p(sub),
                              n_out, _p(out))
    return out


def frames_to_4x4(T) -> np.ndarray:
    T = np.asarray(T)
    out = np.zeros(T.shape[:-1] + (4, 4), np.float32)
    out[..., :3, :3] = T[..., :9].reshape(T.shape[:-1] + (3, 3))
    out[..., :3, 3] = T[..., 9:]
    out[..., 3, 3] = 1
    return out


# ---------------------------------------------------------------- scene clouds (distribution only)
def philox4x32(ctr, key) -> np.ndarray:
    c = np.ascontiguousarray(ctr, dtype=np.uint32)
    k = np.ascontiguousarray(key, dtype=np.uint32)
    out = np.empty(4, np.uint32)
    lib().orc_philox4x32(_p(c), _p(k), _p(out))
    return out


def scene_cloud(scn: dict, num_points: int, seed: int, env_offset: int = 0):
    """Restatement of csrc/scene.hip.  scn: cuboid_{centers,dims,quats}, cylinder_{centers,radii,heights,quats}.
    -> points float32 [B,N,3], assign uint16 [B,N], labels uint8 [B,M1+M2], n_obstacles int32 [B].
    Row b draws as GLOBAL environment ``env_offset + b`` (Philox counter word 1)."""
    cc, cd, cq = _f(scn["cuboid_centers"]), _f(scn["cuboid_dims"]), _f(scn["cuboid_quats"])
    yc, yr, yh, yq = (_f(scn["cylinder_centers"]), _f(scn["cylinder_radii"]), _f(scn["cylinder_heights"]),
                      _f(scn["cylinder_quats"]))
    B, M1 = cd.shape[:2]
    M2 = yr.shape[1]
    assign = np.empty((B, num_points), np.uint16)
    labels = np.zeros((B, M1 + M2), np.uint8)
    nobs = np.zeros(B, np.int32)
    lib().orc_scene_assign(_p(cd), M1, _p(yr), _p(yh), M2, B, num_points, ctypes.c_uint64(seed),
                           ctypes.c_int64(env_offset), _p(assign), _p(labels), _p(nobs))
    pts = np.empty((B, num_points, 3), np.float32)
    lib().orc_scene_points(_p(cc), _p(cd), _p(cq), M1, _p(yc), _p(yr), _p(yh), _p(yq), M2, B, num_points,
                           ctypes.c_uint64(seed), ctypes.c_int64(env_offset), _p(assign), _p(pts))
    return pts, assign, labels, nobs


def success(eff_frames, targets, pos_tol=0.01, cos_tol=float(np.cos(np.radians(15.0)))):
    e, t = _f(eff_frames), _f(targets)
    B = e.shape[0]
    ok = np.zeros(B, np.uint8)
    pe, ca = np.empty(B, np.float32), np.empty(B, np.float32)
    lib().orc_success(_p(e), _p(t), B, ctypes.c_float(pos_tol), ctypes.c_float(cos_tol), _p(ok), _p(pe), _p(ca))
    return ok.astype(bool), pe, ca


def trajectory_metrics(traj, lengths, targets, limits, finger: float = 0.025) -> dict:
    """Row N3: PyBullet-free parts of metrics.py's Evaluator for trajectories [B,T,7]."""
    x, tg, lim = _f(traj), _f(targets), _f(limits)
    B, T = x.shape[:2]
    ln = None if lengths is None else _i(lengths)
    out = {k: np.empty(B, np.float32) for k in ("position_error", "orientation_error", "eff_position_path_length",
                                                "eff_orientation_path_length")}
    jl, sc = np.zeros(B, np.int32), np.zeros(B, np.int32)
    lib().orc_trajectory_metrics(_p(x), _p(ln), _p(tg), _p(lim), B, T, ctypes.c_float(finger), _p(out["position_error"]),
                                 _p(out["orientation_error"]), _p(out["eff_position_path_length"]),
                                 _p(out["eff_orientation_path_length"]), _p(jl), _p(sc))
    out["joint_limit_violation"] = jl.astype(bool)
    out["self_collision"] = sc.astype(bool)
    return out


def sparc(movement, fs: float, padlevel: int = 4, fc: float = 10.0, amp_th: float = 0.05) -> float:
    """Spectral arc length of ONE speed profile, restating ``mpinets/third_party/sparc.py:52-128`` step by step
    (PINNED: ``tests/golden/sparc_golden.npz`` holds that function's own results, incl. its docstring's known answer).

    1. a profile that is all (numerically) zero scores 0 (sparc.py:93-95);
    2. magnitude spectrum of the profile zero-padded to ``2^(ceil(log2 n) + padlevel)`` bins, scaled to a maximum of 1,
       on the frequency axis ``k * fs / nfft`` (sparc.py:97-103) -- the FULL axis up to fs, so with fs < fc the mirrored
       half of the spectrum is part of the curve, as it is for the reference at run_inference's 1 / 0.12 s;
    3. keep the bins with f <= fc (sparc.py:111-113), then the span from the first to the last bin whose magnitude is
       >= amp_th (sparc.py:118-121);
    4. minus the length of the curve (f / (f_last - f_first), magnitude) over that span (sparc.py:124-128)."""
    m = np.asarray(movement, dtype=np.float64)
    if np.allclose(m, 0):
        return 0.0
    nfft = int(2 ** (np.ceil(np.log2(len(m))) + padlevel))
    freq = np.arange(0, fs, fs / nfft)
    mag = np.abs(np.fft.fft(m, nfft))
    mag = mag / mag.max()
    low = np.flatnonzero(freq <= fc)
    freq, mag = freq[low], mag[low]
    loud = np.flatnonzero(mag >= amp_th)
    freq, mag = freq[loud[0]:loud[-1] + 1], mag[loud[0]:loud[-1] + 1]
    steps_f = np.diff(freq) / (freq[-1] - freq[0]) if len(freq) > 1 else np.zeros(0)
    return float(-np.sum(np.sqrt(steps_f ** 2 + np.diff(mag) ** 2)))


def trajectory_smoothness(traj, lengths, dt: float, finger: float = 0.025):
    """``Evaluator.calculate_smoothness`` (mpinets/metrics.py:387-409) for trajectories [B,T,7] with ``lengths`` valid
    waypoints each: SPARC of the joint-space speed profile and of the ``right_gripper`` position's (FK: this oracle's,
    parity unpinned like every FK-derived number).  -> (config_sparc [B], eff_sparc [B]) float64."""
    x = np.asarray(traj, dtype=np.float64)
    B, T = x.shape[:2]
    ln = np.full(B, T, np.int64) if lengths is None else np.asarray(lengths, dtype=np.int64)
    from mpinets_amd import franka_tables as _ft  # (link order of the FK frames: data, not engine code)

    cfg, eff = np.zeros(B), np.zeros(B)
    for b in range(B):
        q = x[b, :ln[b]]
        cfg[b] = sparc(np.linalg.norm(np.diff(q, 1, axis=0) / dt, axis=1), 1.0 / dt)
        frames = franka_fk(q.astype(np.float32), finger)  # [n, links, 12] rows of [R | t]
        pos = frames[:, _ft.LINK_ID["right_gripper"], 9:].astype(np.float64)
        eff[b] = sparc(np.linalg.norm(np.diff(pos, 1, axis=0) / dt, axis=1), 1.0 / dt)
    return cfg, eff


# ---------------------------------------------------------------- depth-camera clouds (row N4)
def depth_render(cam_poses, intr, W, H, cub, cyl, sph_centers=None, sph_radii=None, far_clip: float = 10.0):
    """cam_poses [B,4,4] world-from-camera (OpenGL axes); intr = (fx, fy, cx, cy); cub = (centers, dims, quats),
    cyl = (centers, radii, heights, quats) -> depth [B, H*W] (-1 = no obstacle / robot in front)."""
    cam = _f(cam_poses).reshape(-1, 16)
    B = cam.shape[0]
    cf, cd = prim_frames(cub[0], cub[2]).reshape(B, -1, 12), _f(cub[1]).reshape(B, -1, 3)
    yf = prim_frames(cyl[0], cyl[3]).reshape(B, -1, 12)
    yr, yh = _f(cyl[1]).reshape(B, -1), _f(cyl[2]).reshape(B, -1)
    sc = None if sph_centers is None else _f(sph_centers).reshape(B, -1, 3)
    sr = None if sph_radii is None else _f(sph_radii)
    S = 0 if sc is None else sc.shape[1]
    depth = np.empty((B, W * H), np.float32)
    c = ctypes.c_float
    lib().orc_depth_render(_p(cam), c(intr[0]), c(intr[1]), c(intr[2]), c(intr[3]), W, H, B, _p(cf), _p(cd),
                           cf.shape[1], _p(yf), _p(yr), _p(yh), yf.shape[1], _p(sc), _p(sr), S, c(far_clip), _p(depth))
    return depth


def depth_select(depth, cam_poses, intr, W, H, n_out: int, seed: int, env_offset: int = 0):
    cam = _f(cam_poses).reshape(-1, 16)
    B = cam.shape[0]
    out = np.zeros((B, n_out, 3), np.float32)
    count = np.zeros(B, np.int32)
    c = ctypes.c_float
    lib().orc_depth_select(_p(_f(depth)), _p(cam), c(intr[0]), c(intr[1]), c(intr[2]), c(intr[3]), W, H, B, n_out,
                           ctypes.c_uint32(seed & 0xFFFFFFFF), ctypes.c_uint32((seed >> 32) & 0xFFFFFFFF),
                           ctypes.c_int64(env_offset), _p(out), _p(count))
    return out, count


# ---------------------------------------------------------------- batch assembly (row N2)
def draw_subset(total: int, n_out: int, seed: int, draw: int) -> np.ndarray:
    """Restatement of csrc/franka.hip mpx_draw_subset: n_out of `total` rows, without replacement, in draw order."""
    out = np.empty(n_out, np.int32)
    lib().orc_draw_subset(int(total), int(n_out), ctypes.c_uint32(seed & 0xFFFFFFFF), ctypes.c_uint32((seed >> 32) & 0xFFFFFFFF),
                          ctypes.c_uint32(draw), _p(out))
    return out


def batch_configs(traj, traj_idx, timestep, limits, noise_scale: float = 0.0, seed: int = 0, finger: float = 0.025,
                  sample_offset: int = 0, train: Optional[bool] = None):
    """Restates the joint part of PointCloudBase.get_inputs (data_loader.py:155-185) + the supervision row of
    PointCloudInstanceDataset.__getitem__ (:403-417) for a list of samples; noise = Box-Muller on Philox4x32-10
    blocks (counter (blk, sample_offset + sample, 7, 0), key = seed), the engine's documented generator.  ``train``
    (default: noise_scale > 0): clamp to the limits like every TRAIN sample (data_loader.py:176-178)."""
    train = noise_scale > 0 if train is None else train
    traj, lim = _f(traj), _f(limits)
    L = traj.shape[1]
    B = len(traj_idx)
    q = np.empty((B, 7), np.float32)
    sup = np.empty((B, 7), np.float32)
    fin = np.empty((B, 7), np.float32)
    lo, hi = lim[:, 0], lim[:, 1]
    for b in range(B):
        ti = int(traj_idx[b])
        t = 0 if timestep is None else int(np.clip(timestep[b], 0, L - 1))
        ts = min(t + 1, L - 1)
        v = traj[ti, t].copy()
        if noise_scale > 0:
            z = np.empty(8, np.float32)
            for blk in range(2):
                r = philox4x32([blk, sample_offset + b, 7, 0], [seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF])
                for pr in range(2):
                    u1 = np.float32(1.0) - np.float32(r[2 * pr] >> 8) * np.float32(2.0 ** -24)
                    u2 = np.float32(r[2 * pr + 1] >> 8) * np.float32(2.0 ** -24)
                    rad = np.sqrt(np.float32(-2.0) * np.log(u1, dtype=np.float32), dtype=np.float32)
                    s_, c_ = (float(v[0]) for v in sincos(np.array([np.float32(6.28318530717958647692) * u2], np.float32)))
                    z[4 * blk + 2 * pr], z[4 * blk + 2 * pr + 1] = rad * c_, rad * s_
            v = (np.float32(noise_scale) * z[:7] + v).astype(np.float32)
        if train:
            v = np.minimum(np.maximum(v, lo), hi).astype(np.float32)
        q[b], sup[b], fin[b] = v, traj[ti, ts], traj[ti, L - 1]
    norm = lambda x: ((x - lo) / (hi - lo) * np.float32(2.0) + np.float32(-1.0)).astype(np.float32)
    T = franka_fk(fin, finger)
    pose = frames_to_4x4(T[:, 14])
    return {"q": q, "configuration": norm(q), "supervision": norm(sup), "target_pose": pose,
            "target_position": pose[:, :3, 3].copy()}


# ---------------------------------------------------------------- losses (row N1), torch CPU autograd
def fk_frames_torch(q, finger: float = 0.025):
    """Differentiable restatement of the Franka chain (same public URDF constants as orc_franka_fk):
    q torch [B,7] -> (R [B,15,3,3], t [B,15,3]) in q's dtype.  Used for gradient checks only."""
    import torch

    dt = q.dtype
    JR = [((1, 0, 0), (0, 1, 0), (0, 0, 1)), ((1, 0, 0), (0, 0, 1), (0, -1, 0)), ((1, 0, 0), (0, 0, -1), (0, 1, 0)),
          ((1, 0, 0), (0, 0, -1), (0, 1, 0)), ((1, 0, 0), (0, 0, 1), (0, -1, 0)), ((1, 0, 0), (0, 0, -1), (0, 1, 0)),
          ((1, 0, 0), (0, 0, -1), (0, 1, 0))]
    JT = [(0, 0, 0.333), (0, 0, 0), (0, -0.316, 0), (0.0825, 0, 0), (-0.0825, 0.384, 0), (0, 0, 0), (0.088, 0, 0)]
    B = q.shape[0]
    R = torch.eye(3, dtype=dt).expand(B, 3, 3)
    t = torch.zeros(B, 3, dtype=dt)
    Rs, ts = [R], [t]

    def step(R, t, fr, ft):
        fr = torch.tensor(fr, dtype=dt)
        ft = torch.tensor(ft, dtype=dt)
        return R @ fr, t + R @ ft

    for j in range(7):
        R, t = step(R, t, JR[j], JT[j])
        c, s_ = torch.cos(q[:, j]), torch.sin(q[:, j])
        z, o = torch.zeros_like(c), torch.ones_like(c)
        Rz = torch.stack([torch.stack([c, -s_, z], 1), torch.stack([s_, c, z], 1), torch.stack([z, z, o], 1)], 1)
        R = R @ Rz
        Rs.append(R), ts.append(t)
    I3 = ((1, 0, 0), (0, 1, 0), (0, 0, 1))
    SH = 0.5 ** 0.5
    R8, t8 = step(R, t, I3, (0, 0, 0.107))
    Rh, th = step(R8, t8, ((SH, SH, 0), (-SH, SH, 0), (0, 0, 1)), (0, 0, 0))
    Rl, tl = step(Rh, th, I3, (0, finger, 0.0584))
    Rr, tr = step(Rh, th, I3, (0, -finger, 0.0584))
    Rlt, tlt = step(Rl, tl, I3, (0, 0, 0.045))
    Rrt, trt = step(Rr, tr, I3, (0, 0, 0.045))
    Rg, tg = step(R8, t8, ((-SH, -SH, 0), (SH, -SH, 0), (0, 0, 1)), (0, 0, 0.1))
    for a, b in ((R8, t8), (Rh, th), (Rl, tl), (Rr, tr), (Rlt, tlt), (Rrt, trt), (Rg, tg)):
        Rs.append(a), ts.append(b)
    return torch.stack(Rs, 1), torch.stack(ts, 1)


def robot_cloud_torch(q, table_pts, table_link, subset=None, finger: float = 0.025):
    """Differentiable FrankaSampler.sample: q torch [B,7] -> [B,n,3]."""
    import torch

    R, t = fk_frames_torch(q, finger)
    pts = torch.as_tensor(np.asarray(table_pts), dtype=q.dtype)
    link = torch.as_tensor(np.asarray(table_link)).long()
    if subset is not None:
        sub = torch.as_tensor(np.asarray(subset)).long()
        pts, link = pts[sub], link[sub]
    return torch.einsum("bnij,nj->bni", R[:, link], pts) + t[:, link]


def sdf_torch(points, frames, kind, a, b=None):
    """Differentiable restatement of TorchCuboids.sdf (geometry.py:256-288, kind='cuboid', a=dims [B,M,3]) and
    TorchCylinders.sdf (:478-507, kind='cylinder', a=radii [B,M], b=heights [B,M]); frames [B,M,4,4]."""
    import torch

    proj = torch.einsum("bmij,bnj->bmni", frames[:, :, :3, :3], points) + frames[:, :, None, :3, 3]
    if kind == "cuboid":
        mask = ~(a.abs() <= 1e-8).any(-1)
        d = proj.abs() - (a / 2)[:, :, None, :]
    else:
        mask = ~((a.abs() <= 1e-8) | (b.abs() <= 1e-8))
        rho = torch.linalg.norm(proj[..., :2], dim=-1)
        d = torch.stack((rho.abs() - a[:, :, None], proj[..., 2].abs() - (b / 2)[:, :, None]), -1)
    outside = torch.linalg.norm(torch.maximum(d, torch.zeros_like(d)), dim=-1)
    inside = torch.minimum(d.max(-1).values, torch.zeros_like(outside))
    sdf = torch.where(mask[:, :, None], outside + inside, torch.full_like(outside, float("inf")))
    if sdf.shape[1] == 0:
        return torch.full(points.shape[:2], float("inf"), dtype=points.dtype)
    return sdf.min(1).values


def collision_loss_torch(points, cub_frames, cub_dims, cyl_frames, cyl_radii, cyl_heights, margin: float = 0.03):
    """loss.py:48-95 on torch CPU tensors."""
    import torch

    s = torch.minimum(sdf_torch(points, cub_frames, "cuboid", cub_dims),
                      sdf_torch(points, cyl_frames, "cylinder", cyl_radii, cyl_heights))
    return torch.clamp(margin - s, min=0).mean()


def point_match_loss_torch(a, b):
    """loss.py:31-45."""
    return ((a - b) ** 2).mean() + (a - b).abs().mean()


# ---------------------------------------------------------------- pointnet2_ops (unpinned)
def opt_n_threads(n: int) -> int:
    return int(lib().orc_opt_n_threads(ctypes.c_int(n)))


def set_sqdist_order(order: int) -> None:
    """0 (default): fma(dz,dz,fma(dx,dx,dy*dy)), upstream's expression as LLVM contracts it; 1: the order rounds 1-2
    assumed, fma(dz,dz,fma(dy,dy,dx*dx)).  For the A/B count in tests/test_oracle_pointnet.py only."""
    lib().orc_set_sqdist_order(ctypes.c_int(order))


def get_sqdist_order() -> int:
    return int(lib().orc_get_sqdist_order())


def fps(xyz, npoint: int) -> np.ndarray:
    """xyz [B,N,C>=3] (first 3 columns used) -> int32 [B,npoint]."""
    x = _f(xyz)
    B, N, C = x.shape
    out = np.empty((B, npoint), np.int32)
    lib().orc_fps(_p(x), B, N, C, npoint, _p(out))
    return out


def gather_points(xyz, idx) -> np.ndarray:
    x, ix = _f(xyz), _i(idx)
    B, N, C = x.shape
    out = np.empty((B, ix.shape[1], 3), np.float32)
    lib().orc_gather_points(_p(x), B, N, C, _p(ix), ix.shape[1], _p(out))
    return out


def ball_query(new_xyz, xyz, radius: float, nsample: int, return_counts: bool = False):
    nx, x = _f(new_xyz), _f(xyz)
    B, N, C = x.shape
    npoint = nx.shape[1]
    out = np.empty((B, npoint, nsample), np.int32)
    cnt = np.empty((B, npoint), np.int32)
    lib().orc_ball_query(_p(nx), _p(x), B, N, C, npoint, ctypes.c_float(radius), nsample, _p(out), _p(cnt))
    return (out, cnt) if return_counts else out


def group_points(xyz, new_xyz, feat, idx) -> np.ndarray:
    """-> [B, 3+C, npoint, nsample]  (xyz - centre first, then features)."""
    x, nx, ix = _f(xyz), _f(new_xyz), _i(idx)
    B, N, S = x.shape
    C = 0 if feat is None else feat.shape[1]
    ft = None if feat is None else _f(feat)
    out = np.empty((B, 3 + C, ix.shape[1], ix.shape[2]), np.float32)
    lib().orc_group_points(_p(x), S, _p(nx), _p(ft), _p(ix), B, N, C, ix.shape[1], ix.shape[2], _p(out))
    return out


# ---------------------------------------------------------------- policy network (numpy)
def _linear(x, w, b):
    """float64 accumulate, float32 store."""
    return (x.astype(np.float64) @ w.astype(np.float64).T + b.astype(np.float64)).astype(np.float32)


def _leaky(x, slope=0.01):
    return np.where(x >= 0, x, x * np.float32(slope)).astype(np.float32)


def _group_norm(x, groups, w, b, eps=1e-5):
    B, C = x.shape
    xg = x.astype(np.float64).reshape(B, groups, C // groups)
    mean = xg.mean(axis=2, keepdims=True)
    var = xg.var(axis=2, keepdims=True)  # biased, like torch
    y = ((xg - mean) / np.sqrt(var + eps)).reshape(B, C)
    return (y * w.astype(np.float64) + b.astype(np.float64)).astype(np.float32)


def shared_mlp(x, layers: Sequence[Tuple[np.ndarray, np.ndarray]]) -> np.ndarray:
    """x [..., Cin] -> [..., Cout]; Conv2d 1x1 + ReLU per layer (bn=False)."""
    lead = x.shape[:-1]
    x = x.reshape(-1, x.shape[-1])  # one large product per layer (the BLAS pool sees the whole row count)
    for w, b in layers:
        w2 = w.reshape(w.shape[0], -1)
        x = np.maximum(_linear(x, w2, b), 0).astype(np.float32)
    return x.reshape(lead + (x.shape[-1],))


def sa_module(xyz, feat, npoint, radius, nsample, layers):
    """PointnetSAModule.forward.  xyz [B,N,3]; feat [B,C,N] or None.

    :returns: new_xyz [B,npoint,3] | None, new_feat [B,Cout,npoint], aux dict (indices)
    """
    xyz = _f(xyz)
    B = xyz.shape[0]
    aux = {}
    if npoint is not None:
        fidx = fps(xyz, npoint)
        new_xyz = gather_points(xyz, fidx)
        bidx = ball_query(new_xyz, xyz, radius, nsample)
        grouped = group_points(xyz, new_xyz, feat, bidx)  # [B,3+C,np,ns]
        aux.update(fps_idx=fidx, ball_idx=bidx)
    else:
        new_xyz = None
        g = [np.transpose(xyz, (0, 2, 1))[:, :, None, :]]
        if feat is not None:
            g.append(_f(feat)[:, :, None, :])
        grouped = np.concatenate(g, axis=1)  # [B,3+C,1,N]
    x = np.transpose(grouped, (0, 2, 3, 1))  # [B,np,ns,Cin]
    y = shared_mlp(x, layers)  # [B,np,ns,Cout]
    pooled = y.max(axis=2)  # [B,np,Cout]
    return new_xyz, np.ascontiguousarray(np.transpose(pooled, (0, 2, 1))), aux


def _sa_layers(sd: Dict[str, np.ndarray], i: int):
    return [(sd[f"point_cloud_encoder.SA_modules.{i}.mlps.0.{k}.weight"],
             sd[f"point_cloud_encoder.SA_modules.{i}.mlps.0.{k}.bias"]) for k in (0, 2, 4)]


def break_up_pc(pc) -> Tuple[np.ndarray, np.ndarray]:
    """MPiNetsPointNet._break_up_pc (model.py:395-407): [B,N,M>3] -> xyz [B,N,3], features [B,M-3,N], both contiguous."""
    pc = _f(pc)
    return np.ascontiguousarray(pc[..., 0:3]), np.ascontiguousarray(np.transpose(pc[..., 3:], (0, 2, 1)))


def fc_layer(sd: Dict[str, np.ndarray], x) -> np.ndarray:
    """MPiNetsPointNet.fc_layer (model.py:385-393): Linear, GroupNorm(16), LeakyReLU, Linear, GroupNorm(16), LeakyReLU,
    Linear.  x [B,1024] -> [B,2048]."""
    p = "point_cloud_encoder.fc_layer."
    x = _linear(_f(x), sd[p + "0.weight"], sd[p + "0.bias"])
    x = _leaky(_group_norm(x, 16, sd[p + "1.weight"], sd[p + "1.bias"]))
    x = _linear(x, sd[p + "3.weight"], sd[p + "3.bias"])
    x = _leaky(_group_norm(x, 16, sd[p + "4.weight"], sd[p + "4.bias"]))
    return _linear(x, sd[p + "6.weight"], sd[p + "6.bias"])


def pointnet_encoder(sd: Dict[str, np.ndarray], pc) -> Tuple[np.ndarray, dict]:
    """MPiNetsPointNet.forward (model.py:409-426).  pc [B,N,4] -> [B,2048]."""
    xyz, feat = break_up_pc(pc)
    aux = {}
    xyz1, f1, a1 = sa_module(xyz, feat, 512, 0.05, 128, _sa_layers(sd, 0))
    xyz2, f2, a2 = sa_module(xyz1, f1, 128, 0.3, 128, _sa_layers(sd, 1))
    _, f3, _ = sa_module(xyz2, f2, None, None, None, _sa_layers(sd, 2))
    aux.update(sa1=a1, sa2=a2, xyz1=xyz1, f1=f1, xyz2=xyz2, f2=f2, f3=f3)
    return fc_layer(sd, f3[:, :, 0]), aux


def policy_forward(sd: Dict[str, np.ndarray], pc, q) -> Tuple[np.ndarray, dict]:
    """MotionPolicyNetwork.forward (model.py:75-91)."""
    enc, aux = pointnet_encoder(sd, pc)
    x = _f(q)
    for k in (0, 2, 4, 6, 8):
        x = _linear(x, sd[f"feature_encoder.{k}.weight"], sd[f"feature_encoder.{k}.bias"])
        if k != 8:
            x = _leaky(x)
    x = np.concatenate([enc, x], axis=1)
    for k in (0, 2, 4, 6):
        x = _linear(x, sd[f"decoder.{k}.weight"], sd[f"decoder.{k}.bias"])
        if k != 6:
            x = _leaky(x)
    aux["encoding"] = enc
    return x, aux


def policy_forward_torch(sd, pc, q):
    """Differentiable restatement of MotionPolicyNetwork.forward (model.py:75-91, 409-426) for gradient checks.
    ``sd``: name -> torch CPU tensor (float64 leaves with requires_grad); pc np [B,N,4]; q torch [B,7].
    Sampling / neighbour indices come from the C restatement (they carry no gradient); every neighbourhood
    keeps its nsample padded slots like the reference."""
    import torch
    import torch.nn.functional as F

    pc = _f(pc)
    dt = q.dtype
    B = pc.shape[0]
    bi = torch.arange(B)[:, None]

    def sa(xyz_np, feat, npoint, radius, nsample, i):
        fidx = fps(xyz_np, npoint)
        new_np = gather_points(xyz_np, fidx)
        bidx = torch.as_tensor(ball_query(new_np, xyz_np, radius, nsample)).long()  # [B,np,ns]
        xyz_t, new_t = torch.as_tensor(xyz_np, dtype=dt), torch.as_tensor(new_np, dtype=dt)
        bb = bi[:, :, None]
        h = torch.cat((xyz_t[bb, bidx] - new_t[:, :, None, :], feat[bb, bidx]), dim=-1)
        for k in (0, 2, 4):
            w = sd[f"point_cloud_encoder.SA_modules.{i}.mlps.0.{k}.weight"]
            h = torch.relu(F.linear(h, w.view(w.shape[0], -1), sd[f"point_cloud_encoder.SA_modules.{i}.mlps.0.{k}.bias"]))
        return new_np, h.max(dim=2).values

    xyz = np.ascontiguousarray(pc[..., :3])
    xyz1, f1 = sa(xyz, torch.as_tensor(pc[..., 3:], dtype=dt), 512, 0.05, 128, 0)
    xyz2, f2 = sa(xyz1, f1, 128, 0.3, 128, 1)
    h = torch.cat((torch.as_tensor(xyz2, dtype=dt), f2), dim=-1)
    for k in (0, 2, 4):
        w = sd[f"point_cloud_encoder.SA_modules.2.mlps.0.{k}.weight"]
        h = torch.relu(F.linear(h, w.view(w.shape[0], -1), sd[f"point_cloud_encoder.SA_modules.2.mlps.0.{k}.bias"]))
    x = h.max(dim=1).values
    p = "point_cloud_encoder.fc_layer."
    x = F.linear(x, sd[p + "0.weight"], sd[p + "0.bias"])
    x = F.leaky_relu(F.group_norm(x, 16, sd[p + "1.weight"], sd[p + "1.bias"]))
    x = F.linear(x, sd[p + "3.weight"], sd[p + "3.bias"])
    x = F.leaky_relu(F.group_norm(x, 16, sd[p + "4.weight"], sd[p + "4.bias"]))
    enc = F.linear(x, sd[p + "6.weight"], sd[p + "6.bias"])
    x = q
    for k in (0, 2, 4, 6, 8):
        x = F.linear(x, sd[f"feature_encoder.{k}.weight"], sd[f"feature_encoder.{k}.bias"])
        if k != 8:
            x = F.leaky_relu(x)
    x = torch.cat((enc, x), dim=1)
    for k in (0, 2, 4, 6):
        x = F.linear(x, sd[f"decoder.{k}.weight"], sd[f"decoder.{k}.bias"])
        if k != 6:
            x = F.leaky_relu(x)
    return x


def rollout(sd, xyz, q, steps: int, sampler, limits, unnormalize_out: bool = False):
    """TrainingMotionPolicyNetwork.rollout (model.py:128-183): ``q = clamp(q + f(xyz, q), -1, 1)``; unnormalise; resample
    the robot cloud at the new configuration and overwrite ``xyz[:, :P, :3]`` IN PLACE.  ``sampler(q_unnorm, step)`` ->
    [B,P,3].  Returns the list of steps+1 configurations (normalised, or joint angles when ``unnormalize_out``)."""
    q = _f(q)
    traj = [unnormalize(q, limits) if unnormalize_out else q]
    for i in range(steps):
        dq, _ = policy_forward(sd, xyz, q)
        q = np.clip(q + dq, np.float32(-1), np.float32(1)).astype(np.float32)
        qu = unnormalize(q, limits)
        traj.append(qu if unnormalize_out else q)
        samples = sampler(qu, i)
        xyz[:, : samples.shape[1], :3] = samples
    return traj


def rollout_until_success(sd, q0, target, point_cloud, sampler, limits, max_rollout_length: int = 150):
    """run_inference.py:137-191 for one problem: normalise q0; per step ``q = clamp(q + f(cloud, q), -1, 1)``,
    unnormalise, append, test the end effector against ``target`` [4,4] (closer than 1 cm and 15 degrees: stop, BEFORE
    the cloud is touched), otherwise resample the robot rows of ``point_cloud`` [1,N,4] in place.
    ``sampler(q_unnorm [1,7], step)`` -> [1,P,3].  Returns the trajectory [T,7] (joint angles, start included)."""
    q = _f(q0).reshape(1, 7)
    traj = [q]
    qn = normalize(q, limits)
    tgt = _f(target).reshape(1, 16)
    for i in range(max_rollout_length):
        dq, _ = policy_forward(sd, point_cloud, qn)
        qn = np.clip(qn + dq, np.float32(-1), np.float32(1)).astype(np.float32)
        qt = unnormalize(qn, limits)
        traj.append(qt)
        ok, _, _ = success(franka_fk(qt)[:, RIGHT_GRIPPER_FRAME], tgt)
        if ok[0]:
            break
        samples = sampler(qt, i)
        point_cloud[:, : samples.shape[1], :3] = samples
    return np.concatenate(traj, axis=0)


def repair_quaternions(quats) -> np.ndarray:
    """data_loader.py:202,232: rows whose four components are all ``np.isclose`` to zero get w = 1."""
    q = np.array(quats, copy=True)
    q[np.all(np.isclose(q, 0), axis=-1), 0] = 1
    return q


def validation_reduce(traj, target_position, sphere_table, cub, cyl, finger: float = 0.025):
    """The tail of validation_step (model.py:274-318) for a given unnormalised rollout ``traj`` [B,T,7]: final
    end-effector position error per environment, has_collision [B] by the radius-group reduce (restated as one
    ``sdf <= radius`` test per sphere -- the groups only batch equal radii), and the two averages it returns.
    ``sphere_table`` = (centres, radii, link ids) of FrankaCollisionSampler(with_base_link=False)."""
    traj = _f(traj)
    B, T = traj.shape[:2]
    centres, radii, links = sphere_table
    frames = franka_fk(traj.reshape(-1, 7), finger)
    c = transform_table(frames, centres, links).reshape(B, T, len(radii), 3)
    flags, msdf = collision_flags(c, radii, cub, cyl)
    eff = frames.reshape(B, T, frames.shape[1], 12)[:, -1, RIGHT_GRIPPER_FRAME, 9:]
    err = np.linalg.norm(eff - _f(target_position), axis=1).astype(np.float32)
    margin = (msdf - _f(radii)[None, None, :]).reshape(B, -1).min(axis=1)
    return {"has_collision": flags, "position_error": err, "avg_target_error": np.float32(err.mean()),
            "avg_collision_rate": np.float32(np.count_nonzero(flags) / B), "margin": margin}


def unnormalize(q, limits) -> np.ndarray:
    """utils.py:207-209 with limits (-1, 1)."""
    q = _f(q)
    lim = _f(limits)
    return ((q - np.float32(-1)) * (lim[:, 1] - lim[:, 0]) / np.float32(2) + lim[:, 0]).astype(np.float32)


def normalize(q, limits) -> np.ndarray:
    """utils.py:91-93 with limits (-1, 1)."""
    q = _f(q)
    lim = _f(limits)
    return ((q - lim[:, 0]) / (lim[:, 1] - lim[:, 0]) * np.float32(2) + np.float32(-1)).astype(np.float32)
