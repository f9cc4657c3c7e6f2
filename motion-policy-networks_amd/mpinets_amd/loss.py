"""Training losses of ``mpinets/loss.py`` on the HIP engine (next row N1 of SURVEY.md section 8f).

Same names and signatures as the reference: ``point_match_loss`` (loss.py:31-45), ``collision_loss``
(loss.py:48-95) and ``CollisionAndBCLossContainer`` (loss.py:98-166).  Each is a
``torch.autograd.Function`` whose forward launches one kernel that also writes the analytic gradient
(csrc/loss.hip); the backward is a scale by the incoming scalar.  ``FrankaSampler.sample`` is
differentiable in ``q`` (robot.py) so the container composes exactly like the reference's.
"""
from __future__ import annotations

from typing import Tuple

import torch

from . import _lib, utils
from .geometry import TorchCuboids, TorchCylinders
from .robot import FrankaSampler


class _PointMatch(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input_pc: torch.Tensor, target_pc: torch.Tensor):
        _lib.require_cuda(input_pc, target_pc)
        assert input_pc.shape == target_pc.shape
        a, t = _lib.f32c(input_pc), _lib.f32c(target_pc)
        B = a.size(0) if a.ndim > 1 else 1
        n = a.numel() // max(B, 1)
        sums = torch.empty((B, 2), dtype=torch.float32, device=a.device)
        grad = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        w = 1.0 / max(a.numel(), 1)
        _lib.call("mpx_point_match", _lib.ptr(a), _lib.ptr(t), B, n, w, w, _lib.ptr(sums), _lib.ptr(grad))
        ctx.save_for_backward(grad)
        return sums.sum() * w

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return (None if grad is None else grad * g), None


class _CollisionHinge(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input_pc: torch.Tensor, cuboids: TorchCuboids, cylinders: TorchCylinders, margin: float):
        _lib.require_cuda(input_pc)
        assert input_pc.ndim == 3 and input_pc.size(2) == 3
        p = _lib.f32c(input_pc)
        B, N, _ = p.shape
        M1, M2 = cuboids.centers.size(1), cylinders.centers.size(1)
        sums = torch.empty(B, dtype=torch.float32, device=p.device)
        grad = torch.empty_like(p) if ctx.needs_input_grad[0] else None
        cd = _lib.f32c(cuboids.dims)
        cr, ch = _lib.f32c(cylinders.radii), _lib.f32c(cylinders.heights)
        _lib.call("mpx_collision_hinge", _lib.ptr(p), N * 3, 3, B, N, _lib.ptr(cuboids.inv_frames), _lib.ptr(cd), M1,
                  _lib.ptr(cylinders.inv_frames), _lib.ptr(cr), _lib.ptr(ch), M2, float(margin), _lib.ptr(sums),
                  _lib.ptr(grad), N * 3, 3)
        ctx.save_for_backward(grad)
        ctx.scale = 1.0 / max(B * N, 1)
        return sums.sum() * ctx.scale

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return (None if grad is None else grad * (g * ctx.scale)), None, None, None


def point_match_loss(input_pc: torch.Tensor, target_pc: torch.Tensor) -> torch.Tensor:
    """L2 (mean) + L1 (mean) between two clouds [B,N,3] (loss.py:31-45)."""
    return _PointMatch.apply(input_pc, target_pc)


def collision_loss(input_pc: torch.Tensor, cuboid_centers: torch.Tensor, cuboid_dims: torch.Tensor,
                   cuboid_quaternions: torch.Tensor, cylinder_centers: torch.Tensor, cylinder_radii: torch.Tensor,
                   cylinder_heights: torch.Tensor, cylinder_quaternions: torch.Tensor) -> torch.Tensor:
    """Hinge loss (margin 3 cm) on the signed distance of ``input_pc`` [B,N,3] to the scene (loss.py:48-95)."""
    cuboids = TorchCuboids(cuboid_centers, cuboid_dims, cuboid_quaternions)
    cylinders = TorchCylinders(cylinder_centers, cylinder_radii, cylinder_heights, cylinder_quaternions)
    return _CollisionHinge.apply(input_pc, cuboids, cylinders, 0.03)


class CollisionAndBCLossContainer:
    """Caches the fixed 1024-point robot sampler like the reference container (loss.py:98-166)."""

    def __init__(self):
        self.fk_sampler = None
        self.num_points = 1024

    def __call__(self, input_normalized: torch.Tensor, cuboid_centers: torch.Tensor, cuboid_dims: torch.Tensor,
                 cuboid_quaternions: torch.Tensor, cylinder_centers: torch.Tensor, cylinder_radii: torch.Tensor,
                 cylinder_heights: torch.Tensor, cylinder_quaternions: torch.Tensor,
                 target_normalized: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        if self.fk_sampler is None:
            self.fk_sampler = FrankaSampler(input_normalized.device, num_fixed_points=self.num_points, use_cache=True,
                                            with_base_link=False)
        input_pc = self.fk_sampler.sample(utils.unnormalize_franka_joints(input_normalized))
        target_pc = self.fk_sampler.sample(utils.unnormalize_franka_joints(target_normalized))
        return (
            collision_loss(input_pc, cuboid_centers, cuboid_dims, cuboid_quaternions, cylinder_centers,
                           cylinder_radii, cylinder_heights, cylinder_quaternions),
            point_match_loss(input_pc, target_pc),
        )
