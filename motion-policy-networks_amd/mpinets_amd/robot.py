"""Robot-geometry seam: drop-in equivalents of the robofin classes the reference imports.

Reference call sites (implementations are in un-vendored robofin v0.0.1):
  ``FrankaSampler``            mpinets/model.py:250,267  run_inference.py:64-69,111-116,169,264-265
  ``FrankaCollisionSampler``   mpinets/model.py:268-271,300
  ``FrankaRobot/FrankaRealRobot`` (``JOINT_LIMITS``, ``DOF``, ``fk``)  mpinets/utils.py:50-51,84-85,
                               run_inference.py:176-178, data_loader.py:155-157

All arithmetic runs in ``libmpinets_hip.so`` (csrc/franka.hip).  The kinematic tables are this
repo's (``franka_tables.py``) -- see that file for what is [EXT-RECALL] and what is in-repo data.
"""
from __future__ import annotations

from typing import List, Optional, Tuple, Union

import numpy as np
import torch

from . import _lib
from . import franka_tables as ft


class _SE3Lite:
    """The few SE3 attributes the reference reads from ``FrankaRobot.fk`` (run_inference.py:180-186)."""

    def __init__(self, matrix: np.ndarray):
        self.matrix = np.asarray(matrix, dtype=np.float64)

    @property
    def xyz(self):
        return self.matrix[:3, 3]

    _xyz = xyz

    @property
    def rotation(self):
        return self.matrix[:3, :3]


class FrankaRobot:
    JOINT_LIMITS = ft.JOINT_LIMITS_PUBLISHED
    DOF = ft.DOF
    NEUTRAL = ft.DEFAULT_Q

    @classmethod
    def within_limits(cls, q) -> bool:
        q = np.asarray(q)
        return bool(np.all(q >= cls.JOINT_LIMITS[:, 0]) and np.all(q <= cls.JOINT_LIMITS[:, 1]))

    @staticmethod
    def fk(q, eff_frame: str = "right_gripper", device: Union[str, torch.device] = "cuda:0") -> _SE3Lite:
        """Single-configuration FK on the GPU; returns an SE3-like with ``.matrix`` / ``.xyz``."""
        qt = torch.as_tensor(np.asarray(q, dtype=np.float32)).reshape(1, 7).to(device)
        frames = franka_fk(qt)[0, ft.LINK_ID[eff_frame]].cpu().numpy().astype(np.float64)
        m = np.eye(4)
        m[:3, :3] = frames[:9].reshape(3, 3)
        m[:3, 3] = frames[9:]
        return _SE3Lite(m)


class FrankaRealRobot(FrankaRobot):
    JOINT_LIMITS = ft.JOINT_LIMITS_REAL


def franka_fk(q: torch.Tensor, finger: float = ft.FINGER_OPENING) -> torch.Tensor:
    """q [B,7] -> frames [B,15,12] (R row-major 3x3, then t) for ``franka_tables.LINK_NAMES``."""
    _lib.require_cuda(q)
    assert q.ndim == 2 and q.size(1) == 7
    qc = _lib.f32c(q)
    out = torch.empty((q.size(0), ft.NUM_FRAMES, 12), dtype=torch.float32, device=q.device)
    _lib.call("mpx_franka_fk", _lib.ptr(qc), q.size(0), float(finger), _lib.ptr(out))
    return out


def frames_to_matrix(frames: torch.Tensor) -> torch.Tensor:
    """[...,12] -> [...,4,4]."""
    m = torch.zeros(frames.shape[:-1] + (4, 4), dtype=frames.dtype, device=frames.device)
    m[..., :3, :3] = frames[..., :9].reshape(frames.shape[:-1] + (3, 3))
    m[..., :3, 3] = frames[..., 9:]
    m[..., 3, 3] = 1
    return m


class _SampleFn(torch.autograd.Function):
    """``FrankaSampler.sample`` under autograd (the reference differentiates robofin's torch FK, loss.py:142-147)."""

    @staticmethod
    def forward(ctx, q, sampler, subset, n_out):
        out = torch.empty((q.size(0), n_out, 3), dtype=torch.float32, device=q.device)
        qc = _lib.f32c(q.detach())
        sampler.sample_into(qc, out, subset)
        ctx.sampler, ctx.subset, ctx.n_out = sampler, subset, n_out
        ctx.save_for_backward(qc)
        return out

    @staticmethod
    def backward(ctx, g):
        (qc,) = ctx.saved_tensors
        s, g = ctx.sampler, _lib.f32c(g)
        gq = torch.empty_like(qc)
        _lib.call("mpx_franka_cloud_grad", _lib.ptr(qc), qc.size(0), s.finger, _lib.ptr(s.table_pts),
                  _lib.ptr(s.table_link), _lib.ptr(ctx.subset), ctx.n_out, _lib.ptr(g), g.stride(0), g.stride(1),
                  _lib.ptr(gq))
        return gq, None, None, None


class FrankaSampler:
    """Robot-surface point clouds by FK of a per-link point table.

    ``FrankaSampler(device, num_fixed_points=None, use_cache=False, with_base_link=True)`` as in
    the reference's call sites.  ``sample(q, num_points)`` draws ONE random column subset per call
    from the host ``np.random`` stream and shares it across the batch, like robofin
    (SURVEY.md row a8); ``num_fixed_points`` freezes that subset at construction
    (``loss.py:142-153`` usage).  ``sample_into`` is the engine's zero-copy form used by rollouts.
    """

    def __init__(self, device, num_fixed_points: Optional[int] = None, use_cache: bool = False,
                 with_base_link: bool = True, point_table: Optional[Tuple[np.ndarray, np.ndarray]] = None,
                 finger: float = ft.FINGER_OPENING):
        # ``FrankaSampler("cpu", use_cache=True)`` is how the reference builds its host-side clouds
        # (run_inference.py:262, data_loader.py:101): accepted as a HOST-FACING handle -- inputs and results are CPU
        # tensors, the arithmetic still runs in the HIP library on the current GPU (there is no CPU implementation;
        # without a GPU this raises).
        self.io_device = torch.device(device)
        if self.io_device.type == "cuda":
            self.device = self.io_device
        else:
            if not torch.cuda.is_available():
                raise _lib.MpxError("FrankaSampler needs a GPU (the HIP engine has no CPU fallback)")
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.with_base_link = with_base_link
        self.num_fixed_points = num_fixed_points
        self.finger = float(finger)
        pts, links = point_table if point_table is not None else ft.link_point_table(4096, with_base_link)
        self.table_pts = torch.as_tensor(np.ascontiguousarray(pts, dtype=np.float32)).to(self.device)
        self.table_link = torch.as_tensor(np.ascontiguousarray(links, dtype=np.int32)).to(self.device)
        self.eef_table = torch.as_tensor(ft.end_effector_point_table()).to(self.device)
        self.num_table_points = int(self.table_pts.size(0))
        self._fixed = None
        if num_fixed_points is not None:
            self._fixed = self._draw(num_fixed_points)

    # -- subsets ------------------------------------------------------------------------------
    def _draw(self, n: int, total: Optional[int] = None) -> torch.Tensor:
        total = self.num_table_points if total is None else total
        idx = np.random.choice(total, n, replace=False).astype(np.int32)
        return torch.from_numpy(idx).to(self.device)

    def draw_subset(self, num_points: int) -> torch.Tensor:
        """A device index tensor usable with ``sample_into`` (one host RNG call, no sync later)."""
        return self._draw(num_points)

    # -- reference API ------------------------------------------------------------------------
    def sample(self, q: torch.Tensor, num_points: Optional[int] = None) -> torch.Tensor:
        """q [B,7] (or [7]) joint angles -> [B,P,3]."""
        if q.ndim == 1:
            q = q.unsqueeze(0)
        if self.io_device.type != "cuda":  # host-facing handle: compute on the GPU, hand the result back
            assert not (torch.is_grad_enabled() and q.requires_grad), "differentiable sampling needs the GPU handle"
            q = q.to(self.device, dtype=torch.float32)
        _lib.require_cuda(q)
        if self._fixed is not None:
            subset = self._fixed
        elif num_points is None:
            subset = None
        else:
            subset = self._draw(num_points)
        n_out = self.num_table_points if subset is None else int(subset.numel())
        if torch.is_grad_enabled() and q.requires_grad:
            return _SampleFn.apply(q, self, subset, n_out)
        out = torch.empty((q.size(0), n_out, 3), dtype=torch.float32, device=q.device)
        self.sample_into(q, out, subset)
        return out.to(self.io_device)

    def sample_into(self, q: torch.Tensor, out: torch.Tensor, subset: Optional[torch.Tensor]) -> None:
        """Write ``out[:, :n, :3]`` in place; ``out`` may be the xyz slab itself ([B,N,4] or [B,n,3])."""
        assert q.ndim == 2 and q.size(1) == 7 and out.ndim == 3 and out.size(0) == q.size(0)
        assert out.dtype == torch.float32 and out.stride(2) == 1
        n_out = self.num_table_points if subset is None else int(subset.numel())
        assert out.size(1) >= n_out
        qc = _lib.f32c(q)
        _lib.call("mpx_franka_cloud", _lib.ptr(qc), q.size(0), self.finger, _lib.ptr(self.table_pts),
                  _lib.ptr(self.table_link), _lib.ptr(subset), n_out, _lib.ptr(out), out.stride(0), out.stride(1))

    def sample_end_effector(self, poses: torch.Tensor, num_points: int, frame: str = "right_gripper",
                            subset: Optional[torch.Tensor] = None) -> torch.Tensor:
        """poses [B,4,4] of ``frame`` -> gripper points [B,num_points,3] (run_inference.py:66-69).  ``subset`` (optional,
        int32 indices into the gripper table): use these rows instead of drawing ``num_points`` of them."""
        if poses.ndim == 2:
            poses = poses.unsqueeze(0)
        poses = poses.to(self.device, dtype=torch.float32)  # (no-op for the GPU handle)
        _lib.require_cuda(poses)
        assert poses.shape[1:] == (4, 4)
        table = self.eef_table if frame == "right_gripper" else torch.as_tensor(
            ft.end_effector_point_table(frame=frame)).to(self.device)
        if subset is None:
            subset = self._draw(num_points, total=int(table.size(0)))
        subset = subset.to(device=self.device, dtype=torch.int32).contiguous()
        assert subset.numel() == num_points
        out = torch.empty((poses.size(0), num_points, 3), dtype=torch.float32, device=poses.device)
        pc = _lib.f32c(poses)
        _lib.call("mpx_pose_cloud", _lib.ptr(pc), poses.size(0), _lib.ptr(table), _lib.ptr(subset), num_points,
                  _lib.ptr(out), out.stride(0), out.stride(1))
        return out.to(self.io_device)

    def end_effector_pose(self, q: torch.Tensor, frame: str = "right_gripper") -> torch.Tensor:
        """q [B,7] -> [B,4,4] (model.py:275)."""
        if q.ndim == 1:
            q = q.unsqueeze(0)
        return frames_to_matrix(franka_fk(q.to(self.device, dtype=torch.float32), self.finger)[:, ft.LINK_ID[frame]]).to(self.io_device)


class FrankaCollisionSampler:
    """Collision-sphere model of the arm (model.py:268-271,300)."""

    def __init__(self, device, with_base_link: bool = False, finger: float = ft.FINGER_OPENING):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.MpxError("FrankaCollisionSampler needs a GPU device (no CPU fallback)")
        c, r, l, groups = ft.collision_sphere_table(with_base_link)
        self.centers = torch.from_numpy(c).to(self.device)
        self.radii = torch.from_numpy(r).to(self.device)
        self.links = torch.from_numpy(l).to(self.device)
        self.groups = groups
        self.num_spheres = int(c.shape[0])
        self.finger = float(finger)

    def sphere_centers(self, q: torch.Tensor) -> torch.Tensor:
        """q [B,7] -> [B,S,3] in table order (grouped by radius)."""
        _lib.require_cuda(q)
        qc = _lib.f32c(q)
        out = torch.empty((q.size(0), self.num_spheres, 3), dtype=torch.float32, device=q.device)
        _lib.call("mpx_franka_spheres", _lib.ptr(qc), q.size(0), self.finger, _lib.ptr(self.centers),
                  _lib.ptr(self.links), self.num_spheres, _lib.ptr(out))
        return out

    def compute_spheres(self, q: torch.Tensor) -> List[Tuple[float, torch.Tensor]]:
        """-> [(radius, centres [B,S_r,3]), ...] like robofin's ``compute_spheres``."""
        allc = self.sphere_centers(q)
        return [(r, allc[:, s:s + n]) for r, s, n in self.groups]

    def check(self, q: torch.Tensor, cuboids, cylinders, return_sdf: bool = False):
        """Fused swept-sphere collision check of trajectories (model.py:293-314).

        :param q: [B,T,7] (or [B,7]) joint angles
        :param cuboids: ``geometry.TorchCuboids`` or None;  :param cylinders: ``TorchCylinders`` or None
        :returns: ``has_collision`` bool [B] (and ``min_sdf`` [B,T,S] when ``return_sdf``)
        """
        if q.ndim == 2:
            q = q.unsqueeze(1)
        _lib.require_cuda(q)
        B, T, _ = q.shape
        qc = _lib.f32c(q)
        flags = torch.zeros(B, dtype=torch.int32, device=q.device)
        msdf = torch.empty((B, T, self.num_spheres), dtype=torch.float32, device=q.device) if return_sdf else None
        cf = cd = yf = yr = yh = None
        M1 = M2 = 0
        keep = []
        if cuboids is not None:
            M1 = cuboids.centers.size(1)
            cf, cd = cuboids.inv_frames, _lib.f32c(cuboids.dims)
            keep.append(cd)
        if cylinders is not None:
            M2 = cylinders.centers.size(1)
            yf, yr, yh = cylinders.inv_frames, _lib.f32c(cylinders.radii), _lib.f32c(cylinders.heights)
            keep += [yr, yh]
        _lib.call("mpx_franka_collision", _lib.ptr(qc), B, T, self.finger, _lib.ptr(self.centers),
                  _lib.ptr(self.radii), _lib.ptr(self.links), self.num_spheres, _lib.ptr(cf), _lib.ptr(cd), M1,
                  _lib.ptr(yf), _lib.ptr(yr), _lib.ptr(yh), M2, _lib.ptr(flags), _lib.ptr(msdf))
        has = flags != 0
        return (has, msdf) if return_sdf else has
