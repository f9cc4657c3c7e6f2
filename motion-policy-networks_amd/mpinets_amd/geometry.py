"""Host-side mirror of the reference's ``mpinets/geometry.py`` backed by the HIP engine.

Same class / method names, argument meaning and assertions as the reference
(``/root/reference/mpinets/geometry.py``): ``TorchSpheres`` (:30-123), ``TorchCuboids`` (:126-347),
``TorchCylinders`` (:350-568), ``construct_mixed_point_cloud`` (:571-608).  All arithmetic runs in
``libmpinets_hip.so``; tensors must live on the GPU (there is no CPU fallback).
"""
from __future__ import annotations

import random
from typing import Sequence

import numpy as np
import torch

from . import _lib
from .primitives import Cuboid, Cylinder, Sphere  # noqa: F401  (re-exported for drop-in imports)


def _isclose0(t: torch.Tensor) -> torch.Tensor:
    # torch.isclose(t, 0) with default tolerances == |t| <= 1e-8 (geometry.py:56,155-157,385-388)
    return torch.isclose(t, torch.zeros(1, dtype=t.dtype, device=t.device))


class _FramedPrimitives:
    """Shared constructor work of TorchCuboids / TorchCylinders (geometry.py:151-157, 382-388)."""

    def _init_common(self, centers: torch.Tensor, quaternions: torch.Tensor):
        assert centers.ndim == 3
        assert quaternions.ndim == 3
        _lib.require_cuda(centers, quaternions)
        self.centers = centers
        # It's helpful to ensure the quaternions are normalized (geometry.py:151) -- the engine
        # normalises again inside the frame kernel with its own pinned rounding order.
        self.quats = quaternions / torch.linalg.norm(quaternions, dim=2)[:, :, None]
        self._init_frames(quaternions)

    def _init_frames(self, raw_quats: torch.Tensor):
        B, M, _ = self.centers.shape
        self.inv_frames = torch.empty((B, M, 4, 4), dtype=torch.float32, device=self.centers.device)
        c, q = _lib.f32c(self.centers), _lib.f32c(raw_quats)
        _lib.call("mpx_prim_frames", _lib.ptr(c), _lib.ptr(q), B * M, _lib.ptr(self.inv_frames))

    @staticmethod
    def _flatten(points: torch.Tensor):
        p = _lib.f32c(points)
        return p, p.shape[:-1], int(np.prod(p.shape[1:-1]))


class TorchSpheres(_FramedPrimitives):
    """Batch of M spheres per element; zero-radius spheres are masked (geometry.py:30-123)."""

    def __init__(self, centers: torch.Tensor, radii: torch.Tensor):
        assert centers.ndim == 3
        assert radii.ndim == 3
        assert centers.ndim == radii.ndim
        _lib.require_cuda(centers, radii)
        self.centers = centers
        self.radii = radii
        self.mask = ~_isclose0(self.radii).squeeze(-1)

    def surface_area(self) -> torch.Tensor:
        # kept verbatim in meaning: the reference uses r**3 here (geometry.py:66)
        return 4 * np.pi * torch.pow(self.radii, 3)

    def sample_surface(self, num_points: int) -> torch.Tensor:
        """Samples points from all spheres, including ones with zero volume (geometry.py:69-85): [B, M, num_points, 3].

        Same draw as the reference under the same ``torch.manual_seed``: it calls ``torch.rand`` without a device, i.e.
        on the global CPU generator, normalises the [0,1)^3 samples (the positive octant only -- kept) and scales them.
        Off the hot path (no caller in model / loss / loaders): plain torch on the spheres' device."""
        B, M, _ = self.centers.shape
        unnormalized = torch.rand((B, M, num_points, 3)).to(self.centers.device)
        normalized = unnormalized / torch.linalg.norm(unnormalized, dim=-1)[:, :, :, None]
        return normalized * self.radii[:, :, None, :] + self.centers[:, :, None, :]

    def _sdf(self, points: torch.Tensor) -> torch.Tensor:
        p, oshape, P = self._flatten(points)
        B, M, _ = self.radii.shape
        out = torch.empty((B, P), dtype=torch.float32, device=p.device)
        c, r = _lib.f32c(self.centers), _lib.f32c(self.radii)
        _lib.call("mpx_sphere_sdf", _lib.ptr(c), _lib.ptr(r), B, M, _lib.ptr(p), P, _lib.ptr(out))
        return out.reshape(oshape).type_as(points)

    def sdf(self, points: torch.Tensor) -> torch.Tensor:
        assert points.ndim == 3
        return self._sdf(points)

    def sdf_sequence(self, points: torch.Tensor) -> torch.Tensor:
        assert points.ndim == 4
        return self._sdf(points)


class TorchCuboids(_FramedPrimitives):
    """Batch of M cuboids per element; zero-volume cuboids are masked (geometry.py:126-347)."""

    def __init__(self, centers: torch.Tensor, dims: torch.Tensor, quaternions: torch.Tensor):
        assert dims.ndim == 3
        self.dims = dims
        self._init_common(centers, quaternions)
        self.mask = ~torch.any(_isclose0(self.dims), dim=-1)

    def geometrout(self):
        B, M, _ = self.centers.shape
        mask = self.mask.cpu()
        c, d, q = self.centers.detach().cpu().numpy(), self.dims.detach().cpu().numpy(), self.quats.detach().cpu().numpy()
        return [[Cuboid(center=c[b, m], dims=d[b, m], quaternion=q[b, m]) for m in range(M) if mask[b, m]]
                for b in range(B)]

    def surface_area(self) -> torch.Tensor:
        d = self.dims
        return 2 * (d[:, :, 0] * d[:, :, 1] + d[:, :, 0] * d[:, :, 2] + d[:, :, 1] * d[:, :, 2])

    def _sdf(self, points: torch.Tensor) -> torch.Tensor:
        assert points.size(0) == self.centers.size(0)
        _lib.require_cuda(points)
        p, oshape, P = self._flatten(points)
        B, M, _ = self.centers.shape
        out = torch.empty((B, P), dtype=torch.float32, device=p.device)
        d = _lib.f32c(self.dims)
        _lib.call("mpx_cuboid_sdf", _lib.ptr(self.inv_frames), _lib.ptr(d), B, M, _lib.ptr(p), P, _lib.ptr(out))
        return out.reshape(oshape).type_as(points)

    def sdf(self, points: torch.Tensor) -> torch.Tensor:
        """points [B,N,3] -> [B,N]: min over the unmasked cuboids, +inf if none (geometry.py:238-288)."""
        assert points.ndim == 3
        return self._sdf(points)

    def sdf_sequence(self, points: torch.Tensor) -> torch.Tensor:
        """points [B,T,N,3] -> [B,T,N] (geometry.py:290-347)."""
        assert points.ndim == 4
        return self._sdf(points)


class TorchCylinders(_FramedPrimitives):
    """Batch of M cylinders per element; zero radius/height are masked (geometry.py:350-568)."""

    def __init__(self, centers: torch.Tensor, radii: torch.Tensor, heights: torch.Tensor,
                 quaternions: torch.Tensor):
        assert radii.ndim == 3
        assert heights.ndim == 3
        self.radii = radii
        self.heights = heights
        self._init_common(centers, quaternions)
        self.mask = ~torch.logical_or(_isclose0(self.radii).squeeze(-1), _isclose0(self.heights).squeeze(-1))

    def geometrout(self):
        B, M, _ = self.centers.shape
        mask = self.mask.cpu()
        c, q = self.centers.detach().cpu().numpy(), self.quats.detach().cpu().numpy()
        r, h = self.radii.detach().cpu().numpy(), self.heights.detach().cpu().numpy()
        return [[Cylinder(center=c[b, m], radius=r[b, m, 0], height=h[b, m, 0], quaternion=q[b, m])
                 for m in range(M) if mask[b, m]] for b in range(B)]

    def _sdf(self, points: torch.Tensor) -> torch.Tensor:
        assert points.size(0) == self.centers.size(0)
        _lib.require_cuda(points)
        p, oshape, P = self._flatten(points)
        B, M, _ = self.centers.shape
        out = torch.empty((B, P), dtype=torch.float32, device=p.device)
        r, h = _lib.f32c(self.radii), _lib.f32c(self.heights)
        _lib.call("mpx_cylinder_sdf", _lib.ptr(self.inv_frames), _lib.ptr(r), _lib.ptr(h), B, M, _lib.ptr(p), P,
                  _lib.ptr(out))
        return out.reshape(oshape).type_as(points)

    def sdf(self, points: torch.Tensor) -> torch.Tensor:
        assert points.ndim == 3
        return self._sdf(points)

    def sdf_sequence(self, points: torch.Tensor) -> torch.Tensor:
        assert points.ndim == 4
        return self._sdf(points)


def construct_mixed_point_cloud(obstacles: Sequence, num_points: int) -> np.ndarray:
    """Random scene point cloud, points allotted by surface area (geometry.py:571-608).

    Host API kept for ``run_inference.py:110`` / ``data_loader.py:258``: any object with
    ``.surface_area`` and ``.sample_surface(n)`` works.  Consumes the ``random`` / ``np.random``
    global streams in the reference's order (label shuffle, per-obstacle samples, final
    ``np.random.choice``), so with identical obstacle samplers and seeds the result is identical.
    Returns float64 ``[num_points, 4]`` = (x, y, z, label in 1..K); an empty list gives shape (1, 0).
    The batched on-device sampler is ``mpinets_amd.scene.sample_scene_clouds``.
    """
    count = len(obstacles)
    if count == 0:
        return np.array([[]])
    areas = np.array([o.surface_area for o in obstacles])
    shares = (areas / np.sum(areas)).tolist()
    labels = list(range(1, count + 1))
    random.shuffle(labels)
    chunks = []
    for label, (obstacle, share) in zip(labels, zip(obstacles, shares)):
        n = int(share * num_points) + 500
        block = np.full((n, 4), float(label))
        block[:, :3] = obstacle.sample_surface(n)
        chunks.append(block)
    pool = np.concatenate(chunks, axis=0)
    return pool[np.random.choice(pool.shape[0], num_points, replace=False), :]
