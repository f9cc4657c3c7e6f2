"""NumPy scene primitives with the duck-typed surface the reference expects.

The reference takes ``Cuboid`` / ``Cylinder`` / ``Sphere`` from the un-vendored ``geometrout``
0.0.3.4 (``/root/reference/docker/Dockerfile:142``) and only relies on ``.surface_area``,
``.sample_surface(n)`` and ``.is_zero_volume()`` on the hot path
(``mpinets/geometry.py:590,600``, ``mpinets/data_loader.py:237-256``).  These are the engine's
own host-side equivalents; quaternions are (w, x, y, z) and rotations are proper rotations.
"""
from __future__ import annotations

import numpy as np


def quat_to_matrix(q) -> np.ndarray:
    w, x, y, z = np.asarray(q, dtype=np.float64) / np.linalg.norm(q)
    return np.array(
        [
            [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
            [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
            [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)],
        ]
    )


class _Posed:
    def __init__(self, center, quaternion):
        self.center = np.asarray(center, dtype=np.float64).reshape(3)
        self.quaternion = np.asarray(quaternion, dtype=np.float64).reshape(4)
        self._R = quat_to_matrix(self.quaternion)

    def _to_world(self, local: np.ndarray) -> np.ndarray:
        return local @ self._R.T + self.center

    def _to_local(self, world: np.ndarray) -> np.ndarray:
        return (np.asarray(world, dtype=np.float64) - self.center) @ self._R


class Cuboid(_Posed):
    def __init__(self, center, dims, quaternion=(1.0, 0.0, 0.0, 0.0)):
        super().__init__(center, quaternion)
        self.dims = np.asarray(dims, dtype=np.float64).reshape(3)

    @property
    def half_extents(self):
        return self.dims / 2

    @property
    def surface_area(self) -> float:
        x, y, z = self.dims
        return float(2 * (x * y + x * z + y * z))

    def is_zero_volume(self) -> bool:
        return bool(np.isclose(self.dims, 0).any())

    def sample_surface(self, num_points: int) -> np.ndarray:
        """Uniform samples on the six faces (face chosen in proportion to its area)."""
        x, y, z = self.dims
        face_area = np.array([y * z, y * z, x * z, x * z, x * y, x * y])
        face = np.random.choice(6, size=num_points, p=face_area / face_area.sum())
        pts = np.random.uniform(-0.5, 0.5, (num_points, 3)) * self.dims
        axis = face // 2
        sign = np.where(face % 2 == 0, -0.5, 0.5)
        pts[np.arange(num_points), axis] = sign * self.dims[axis]
        return self._to_world(pts)

    def sdf(self, points) -> np.ndarray:
        p = self._to_local(points)
        d = np.abs(p) - self.half_extents
        return np.linalg.norm(np.maximum(d, 0), axis=-1) + np.minimum(d.max(axis=-1), 0)


class Cylinder(_Posed):
    def __init__(self, center, radius, height, quaternion=(1.0, 0.0, 0.0, 0.0)):
        super().__init__(center, quaternion)
        self.radius = float(radius)
        self.height = float(height)

    @property
    def surface_area(self) -> float:
        return float(2 * np.pi * self.radius * self.height + 2 * np.pi * self.radius**2)

    def is_zero_volume(self) -> bool:
        return bool(np.isclose(self.radius, 0) or np.isclose(self.height, 0))

    def sample_surface(self, num_points: int) -> np.ndarray:
        side = 2 * np.pi * self.radius * self.height
        cap = np.pi * self.radius**2
        which = np.random.choice(3, size=num_points, p=np.array([side, cap, cap]) / (side + 2 * cap))
        theta = np.random.uniform(0, 2 * np.pi, num_points)
        rho = np.where(which == 0, self.radius, self.radius * np.sqrt(np.random.uniform(0, 1, num_points)))
        zz = np.where(which == 0, np.random.uniform(-0.5, 0.5, num_points) * self.height,
                      np.where(which == 1, -0.5 * self.height, 0.5 * self.height))
        pts = np.stack([rho * np.cos(theta), rho * np.sin(theta), zz], axis=1)
        return self._to_world(pts)

    def sdf(self, points) -> np.ndarray:
        p = self._to_local(points)
        d = np.stack([np.linalg.norm(p[..., :2], axis=-1) - self.radius, np.abs(p[..., 2]) - self.height / 2], -1)
        return np.linalg.norm(np.maximum(d, 0), axis=-1) + np.minimum(d.max(axis=-1), 0)


class Sphere:
    def __init__(self, center, radius):
        self.center = np.asarray(center, dtype=np.float64).reshape(3)
        self.radius = float(radius)

    @property
    def surface_area(self) -> float:
        return float(4 * np.pi * self.radius**2)

    def is_zero_volume(self) -> bool:
        return bool(np.isclose(self.radius, 0))

    def sample_surface(self, num_points: int) -> np.ndarray:
        v = np.random.normal(size=(num_points, 3))
        return self.center + self.radius * v / np.linalg.norm(v, axis=1, keepdims=True)

    def sdf(self, points) -> np.ndarray:
        return np.linalg.norm(np.asarray(points, dtype=np.float64) - self.center, axis=-1) - self.radius
