"""Partial-view scene clouds from a depth camera (next row N4 of SURVEY.md section 8f).

The reference converts primitive problems to depth clouds one at a time with PyBullet
(``run_inference.py:194-257`` ``convert_primitive_problems_to_depth``: fixed camera pose per environment type,
``sim.get_pointcloud_from_camera(camera, remove_robot=franka)``) and later draws 4096 of the points
(``run_inference.py:78-85``).  Here a whole batch is ray-cast analytically on the GPU (``mpx_depth_render``) and
the random subset is drawn on the device (``mpx_depth_select``).  PyBullet's rasteriser, its depth quantisation and
robofin's camera intrinsics are not in the tree: **parity unpinned**; geometry is exact for the primitives.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .geometry import TorchCuboids, TorchCylinders
from .robot import FrankaCollisionSampler

# world-from-camera poses used for the paper's evaluations (run_inference.py:215-243): xyz + wxyz quaternion
EVAL_CAMERAS = {
    "dresser": ([0.08307640315968651, 1.986952324350807, 0.9996085854670145],
                [-0.10162310189063647, -0.06726290364234049, 0.5478233048853433, 0.8276702686337273]),
    "cubby": ([0.08307640315968651, 1.986952324350807, 0.9996085854670145],
              [-0.10162310189063647, -0.06726290364234049, 0.5478233048853433, 0.8276702686337273]),
    "tabletop": ([1.5031788593125708, -1.817341016921562, 1.278088299149147],
                 [0.8687241016192855, 0.4180885960330695, 0.11516106409944685, 0.23928704613569252]),
}


def camera_pose(environment_type: str) -> np.ndarray:
    """4x4 world-from-camera matrix of the evaluation camera whose name occurs in ``environment_type``."""
    for name, (xyz, q) in EVAL_CAMERAS.items():
        if name in environment_type:
            w, x, y, z = np.asarray(q, dtype=np.float64) / np.linalg.norm(q)
            R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                          [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                          [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
            m = np.eye(4)
            m[:3, :3], m[:3, 3] = R, xyz
            return m.astype(np.float32)
    raise NotImplementedError(f"Camera angle is not implemented for environment type: {environment_type}")


class DepthCamera:
    """Pinhole camera, OpenGL axes.  Defaults: 640 x 480, 60 degree vertical field of view ([EXT-RECALL]: robofin's
    PyBullet camera parameters are not in the tree)."""

    def __init__(self, width: int = 640, height: int = 480, vertical_fov_deg: float = 60.0, far_clip: float = 10.0):
        self.width, self.height, self.far_clip = int(width), int(height), float(far_clip)
        self.fy = 0.5 * height / math.tan(math.radians(vertical_fov_deg) / 2)
        self.fx = self.fy
        self.cx, self.cy = width / 2.0, height / 2.0

    @property
    def intrinsics(self):
        return (self.fx, self.fy, self.cx, self.cy)

    @torch.no_grad()
    def render(self, cam_poses: torch.Tensor, cuboids: TorchCuboids, cylinders: TorchCylinders,
               q: Optional[torch.Tensor] = None, collision_sampler: Optional[FrankaCollisionSampler] = None) -> torch.Tensor:
        """-> depth [B, H, W] (metres along the ray; -1 = nothing / robot).  ``q`` [B,7]: robot pixels are removed."""
        _lib.require_cuda(cam_poses)
        cam = _lib.f32c(cam_poses)
        B = cam.size(0)
        sc, sr, S = None, None, 0
        if q is not None:
            cs = collision_sampler or FrankaCollisionSampler(cam.device, with_base_link=True)
            sc, sr, S = cs.sphere_centers(q).contiguous(), cs.radii, cs.num_spheres
        depth = torch.empty((B, self.height, self.width), dtype=torch.float32, device=cam.device)
        cd, yr, yh = _lib.f32c(cuboids.dims), _lib.f32c(cylinders.radii), _lib.f32c(cylinders.heights)
        _lib.call("mpx_depth_render", _lib.ptr(cam), self.fx, self.fy, self.cx, self.cy, self.width, self.height, B,
                  _lib.ptr(cuboids.inv_frames), _lib.ptr(cd), cuboids.centers.size(1), _lib.ptr(cylinders.inv_frames),
                  _lib.ptr(yr), _lib.ptr(yh), cylinders.centers.size(1), _lib.ptr(sc), _lib.ptr(sr), S, self.far_clip,
                  _lib.ptr(depth))
        return depth

    @torch.no_grad()
    def sample_cloud(self, depth: torch.Tensor, cam_poses: torch.Tensor, num_points: int, seed: int = 0,
                     out: Optional[torch.Tensor] = None, env_offset: int = 0) -> torch.Tensor:
        """``num_points`` of the valid pixels, uniformly without replacement, as world points [B,num_points,3]
        (``out`` may be slab rows ``xyz[:, 2048:6144]``).  Raises ValueError like ``np.random.choice`` if an
        image has fewer valid pixels.  Row b draws as global environment ``env_offset + b`` (sharded batches)."""
        cam = _lib.f32c(cam_poses)
        B = cam.size(0)
        if out is None:
            out = torch.empty((B, num_points, 3), dtype=torch.float32, device=cam.device)
        assert out.size(0) == B and out.size(1) >= num_points and out.stride(2) == 1
        count = torch.empty(B, dtype=torch.int32, device=cam.device)
        d = _lib.f32c(depth)
        _lib.call("mpx_depth_select", _lib.ptr(d), _lib.ptr(cam), self.fx, self.fy, self.cx, self.cy, self.width,
                  self.height, B, num_points, int(seed) & (2 ** 64 - 1), int(env_offset), _lib.ptr(out), out.stride(0),
                  out.stride(1),
                  _lib.ptr(count))
        self.last_counts = count
        if int(count.min().item()) < num_points:
            raise ValueError("Cannot take a larger sample than population when 'replace=False' "
                             f"(an image has {int(count.min().item())} valid pixels, {num_points} requested)")
        return out


def depth_point_clouds(prims: Dict[str, torch.Tensor], q0: torch.Tensor, environment_types: Sequence[str],
                       num_points: int = 4096, seed: int = 0, camera: Optional[DepthCamera] = None,
                       out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Batched ``convert_primitive_problems_to_depth`` + the 4096-point draw: primitives (dict of the usual seven
    arrays on the GPU), start configurations [B,7], one environment type per problem -> [B,num_points,3]."""
    camera = camera or DepthCamera()
    dev = q0.device
    poses = torch.from_numpy(np.stack([camera_pose(t) for t in environment_types])).to(dev)
    cub = TorchCuboids(prims["cuboid_centers"], prims["cuboid_dims"], prims["cuboid_quats"])
    cyl = TorchCylinders(prims["cylinder_centers"], prims["cylinder_radii"], prims["cylinder_heights"],
                         prims["cylinder_quats"])
    depth = camera.render(poses, cub, cyl, q=q0)
    return camera.sample_cloud(depth, poses, num_points, seed, out=out)
