"""Closed-loop batched rollout engine: the per-step hot path with no host synchronisation.

One ``step()`` = policy forward -> ``q = clamp(q + dq, -1, 1)`` -> unnormalise -> FK + robot-cloud
refresh written in place into ``xyz[:, :P, :3]`` -> swept-sphere SDF collision check of the new
configuration.  This is ``TrainingMotionPolicyNetwork.rollout`` (mpinets/model.py:170-181) plus the
per-waypoint part of the validation collision sweep (model.py:293-314) for a batch of independent
planning problems; the reference's inference loop does the same for B = 1 with a device->host
sync every step (run_inference.py:171-189), which this engine does not need.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch

from . import _lib
from . import franka_tables as ft
from .geometry import TorchCuboids, TorchCylinders
from .model import MotionPolicyNetwork
from .robot import FrankaCollisionSampler, FrankaSampler


class RolloutEngine:
    def __init__(self, model: MotionPolicyNetwork, problem: Dict[str, torch.Tensor], num_robot_points: int = 2048,
                 robot_subset: Optional[torch.Tensor] = None):
        self.model = model
        dev = problem["xyz"].device
        self.device = dev
        self.xyz = problem["xyz"]
        self.q_norm = _lib.f32c(problem["q_norm"]).clone()
        self.q = torch.empty_like(self.q_norm)
        self.B = self.xyz.size(0)
        self.sampler = FrankaSampler(dev)
        self.subset = robot_subset if robot_subset is not None else problem.get("robot_subset")
        if self.subset is None:
            self.subset = self.sampler.draw_subset(num_robot_points)
        self.collision = FrankaCollisionSampler(dev, with_base_link=False)
        self.cuboids = TorchCuboids(problem["cuboid_centers"], problem["cuboid_dims"], problem["cuboid_quats"])
        self.cylinders = TorchCylinders(problem["cylinder_centers"], problem["cylinder_radii"],
                                        problem["cylinder_heights"], problem["cylinder_quats"])
        self._cd = _lib.f32c(self.cuboids.dims)
        self._yr, self._yh = _lib.f32c(self.cylinders.radii), _lib.f32c(self.cylinders.heights)
        self.limits = torch.as_tensor(ft.JOINT_LIMITS_REAL, dtype=torch.float32, device=dev).contiguous()
        self.flags = torch.zeros(self.B, dtype=torch.int32, device=dev)
        self.steps_done = 0

    @torch.no_grad()
    def step(self) -> torch.Tensor:
        """Advance every environment by one policy step; returns the new joint angles [B,7]."""
        lib = _lib
        dq = self.model(self.xyz, self.q_norm)
        lib.call("mpx_joint_step", lib.ptr(self.q_norm), lib.ptr(dq), lib.ptr(self.limits), self.B,
                 lib.ptr(self.q_norm), lib.ptr(self.q))
        self.sampler.sample_into(self.q, self.xyz, self.subset)
        c = self.collision
        lib.call("mpx_franka_collision", lib.ptr(self.q), self.B, 1, c.finger, lib.ptr(c.centers),
                 lib.ptr(c.radii), lib.ptr(c.links), c.num_spheres, lib.ptr(self.cuboids.inv_frames),
                 lib.ptr(self._cd), self.cuboids.centers.size(1), lib.ptr(self.cylinders.inv_frames),
                 lib.ptr(self._yr), lib.ptr(self._yh), self.cylinders.centers.size(1), lib.ptr(self.flags), None)
        self.steps_done += 1
        return self.q

    def rollout(self, steps: int) -> torch.Tensor:
        """-> trajectory [B, steps+1, 7] (joint space), like ``rollout(..., unnormalize=True)``."""
        lim = self.limits
        q0 = (self.q_norm + 1) * (lim[:, 1] - lim[:, 0]) / 2 + lim[:, 0]
        traj = [q0]
        for _ in range(steps):
            traj.append(self.step().clone())
        return torch.stack(traj, dim=1)

    @property
    def has_collision(self) -> torch.Tensor:
        return self.flags != 0
