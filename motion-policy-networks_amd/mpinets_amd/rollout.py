"""Closed-loop batched rollout engine: the per-step hot path with no host synchronisation.

One ``step()`` = policy forward -> ``q = clamp(q + dq, -1, 1)`` -> unnormalise -> FK + robot-cloud
refresh written in place into ``xyz[:, :P, :3]`` -> swept-sphere SDF collision check of the new
configuration.  This is ``TrainingMotionPolicyNetwork.rollout`` (mpinets/model.py:170-181) plus the
per-waypoint part of the validation collision sweep (model.py:293-314) for a batch of independent
planning problems; the reference's inference loop does the same for B = 1 with a device->host
sync every step (run_inference.py:171-189), which this engine does not need.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from . import franka_tables as ft
from .geometry import TorchCuboids, TorchCylinders
from .model import MotionPolicyNetwork
from .robot import FrankaCollisionSampler, FrankaSampler


class RolloutEngine:
    def __init__(self, model: MotionPolicyNetwork, problem: Dict[str, torch.Tensor], num_robot_points: int = 2048,
                 robot_subset: Optional[torch.Tensor] = None, rerender_scene: bool = False, scene_seed: int = 0,
                 env_offset: Optional[int] = None, resample_subset: bool = False, subset_seed: int = 0):
        """``rerender_scene``: draw a fresh 4096-point scene cloud from the primitives at the start of every step
        (BASELINE config 5, "closed-loop point-cloud re-render"); default keeps the scene rows of the slab.
        ``env_offset``: global id of this batch's first environment (default: the problem's own ``env_offset`` entry,
        else 0).  The re-render draws are keyed by (``scene_seed``, step, global environment id), so a rank that owns
        environments [o, o+B) of a sharded batch computes exactly what a single process computes for those rows.
        ``resample_subset``: redraw the robot cloud's point subset at EVERY step, one draw shared by the batch -- what the
        reference's loop does (robofin's ``FrankaSampler.sample`` draws ``np.random.choice`` per call: model.py:170-181,
        run_inference.py:188-189).  The draw happens on the device (``mpx_draw_subset``, keyed by (``subset_seed``,
        step): no host RNG, no synchronisation, the same subset on every rank of a sharded batch).  Default ``False``:
        one subset for the whole rollout (a fixed robot sampling pattern: the same distribution at every single step,
        bit-reproducible FPS indices from step to step)."""
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
        self.resample_subset, self.subset_seed = bool(resample_subset), int(subset_seed)
        if self.resample_subset:  # own buffer: rewritten every step
            self.subset = self.subset.to(device=dev, dtype=torch.int32).clone().contiguous()
            assert self.subset.numel() <= self.sampler.table_pts.size(0)
        self.collision = FrankaCollisionSampler(dev, with_base_link=False)
        self.cuboids = TorchCuboids(problem["cuboid_centers"], problem["cuboid_dims"], problem["cuboid_quats"])
        self.cylinders = TorchCylinders(problem["cylinder_centers"], problem["cylinder_radii"],
                                        problem["cylinder_heights"], problem["cylinder_quats"])
        self._cd = _lib.f32c(self.cuboids.dims)
        self._yr, self._yh = _lib.f32c(self.cylinders.radii), _lib.f32c(self.cylinders.heights)
        self.limits = torch.as_tensor(ft.JOINT_LIMITS_REAL, dtype=torch.float32, device=dev).contiguous()
        self.flags = torch.zeros(self.B, dtype=torch.int32, device=dev)
        self.steps_done = 0
        self.rerender_scene, self.scene_seed = bool(rerender_scene), int(scene_seed)
        self.env_offset = int(problem.get("env_offset", 0) if env_offset is None else env_offset)
        if self.rerender_scene:
            self._prims = {k: problem[k] for k in ("cuboid_centers", "cuboid_dims", "cuboid_quats", "cylinder_centers",
                                                   "cylinder_radii", "cylinder_heights", "cylinder_quats")}
            self._n_robot = int(self.subset.numel())
            self._n_scene = self.xyz.size(1) - self._n_robot - 128
            m = self._prims["cuboid_dims"].size(1) + self._prims["cylinder_radii"].size(1)
            self._scene_scratch = (torch.empty((self.B, self._n_scene), dtype=torch.int16, device=dev),
                                   torch.zeros((self.B, m), dtype=torch.uint8, device=dev),
                                   torch.zeros(self.B, dtype=torch.int32, device=dev))
        # tests: a dict here receives the forward's intermediates (FPS / ball-query indices, counts, features)
        self.capture: Optional[dict] = None
        # success tracking (rollout_until_success): target poses, done flags, per-env step counts
        self.targets = None
        self.done = None
        self.steps = None

    def track_success(self, target_poses: torch.Tensor, pos_tol: float = 0.01, rot_tol_deg: float = 15.0):
        """Enable the on-device early-stop test of run_inference.py:180-187 against ``target_poses`` [B,4,4]
        (``right_gripper`` frame).  Finished environments keep their last configuration."""
        assert target_poses.shape == (self.B, 4, 4)
        self.targets = _lib.f32c(target_poses)
        self.done = torch.zeros(self.B, dtype=torch.int32, device=self.device)
        self.steps = torch.zeros(self.B, dtype=torch.int32, device=self.device)
        self.pos_tol = float(pos_tol)
        self.cos_tol = float(np.cos(np.radians(rot_tol_deg)))

    @torch.no_grad()
    def step(self) -> torch.Tensor:
        """Advance every environment by one policy step; returns the new joint angles [B,7].

        (A HIP-graph replay of the step was measured and gives nothing: at B = 1 the eager step takes 2.40 ms and
        the captured one 2.40 ms -- the time is the 638 dependent farthest-point picks and kernel tails, not launch
        overhead -- so the step stays a plain sequence of launches on the caller's stream.)"""
        return self._step_eager()

    @torch.no_grad()
    def _step_eager(self) -> torch.Tensor:
        lib = _lib
        if self.rerender_scene:
            from .scenes import sample_scene_clouds

            sample_scene_clouds(self._prims, self._n_scene, self.scene_seed + 7919 * self.steps_done,
                                out=self.xyz[:, self._n_robot:self._n_robot + self._n_scene],
                                scratch=self._scene_scratch, env_offset=self.env_offset)
        dq = self.model(self.xyz, self.q_norm) if self.capture is None else self.model(self.xyz, self.q_norm, aux=self.capture)
        lib.call("mpx_joint_step", lib.ptr(self.q_norm), lib.ptr(dq), lib.ptr(self.limits), self.B,
                 lib.ptr(self.q_norm), lib.ptr(self.q), lib.ptr(self.done))
        if self.done is not None:
            lib.call("mpx_franka_success", lib.ptr(self.q), lib.ptr(self.targets), self.B, self.sampler.finger,
                     self.pos_tol, self.cos_tol, lib.ptr(self.done), lib.ptr(self.steps), None, None)
        if self.resample_subset:
            lib.call("mpx_draw_subset", self.sampler.table_pts.size(0), self.subset.numel(), self.subset_seed & (2 ** 64 - 1),
                     self.steps_done, lib.ptr(self.subset))
        self.sampler.sample_into(self.q, self.xyz, self.subset)
        c = self.collision
        lib.call("mpx_franka_collision", lib.ptr(self.q), self.B, 1, c.finger, lib.ptr(c.centers),
                 lib.ptr(c.radii), lib.ptr(c.links), c.num_spheres, lib.ptr(self.cuboids.inv_frames),
                 lib.ptr(self._cd), self.cuboids.centers.size(1), lib.ptr(self.cylinders.inv_frames),
                 lib.ptr(self._yr), lib.ptr(self._yh), self.cylinders.centers.size(1), lib.ptr(self.flags), None)
        self.steps_done += 1
        self.last_counts = getattr(self.model.point_cloud_encoder, "last_counts", None)  # this share's hit counts
        return self.q

    # ---- the same step through the single C entry point (mpx_rollout_step): what a caller without Python uses ----
    class _NativeScene(ctypes.Structure):  # field order = struct mpx_rollout_scene (include/mpinets_hip.h)
        _fields_ = [("limits", ctypes.c_void_p), ("finger", ctypes.c_float), ("table_pts", ctypes.c_void_p),
                    ("table_link", ctypes.c_void_p), ("subset", ctypes.c_void_p), ("n_robot", ctypes.c_int),
                    ("sph_centers", ctypes.c_void_p), ("sph_radii", ctypes.c_void_p), ("sph_link", ctypes.c_void_p),
                    ("n_spheres", ctypes.c_int), ("cub_frames", ctypes.c_void_p), ("cub_dims", ctypes.c_void_p),
                    ("M1", ctypes.c_int), ("cyl_frames", ctypes.c_void_p), ("cyl_radii", ctypes.c_void_p),
                    ("cyl_heights", ctypes.c_void_p), ("M2", ctypes.c_int)]

    class _NativeOptions(ctypes.Structure):  # field order = struct mpx_rollout_options (include/mpinets_hip.h)
        _fields_ = [("steps", ctypes.c_int), ("first_step", ctypes.c_int), ("n_scene", ctypes.c_int),
                    ("scene_seed", ctypes.c_uint64), ("env_offset", ctypes.c_int64), ("cub_centers", ctypes.c_void_p),
                    ("cub_quats", ctypes.c_void_p), ("cyl_centers", ctypes.c_void_p), ("cyl_quats", ctypes.c_void_p),
                    ("target_poses", ctypes.c_void_p), ("pos_tol", ctypes.c_float), ("cos_rot_tol", ctypes.c_float),
                    ("done", ctypes.c_void_p), ("steps_taken", ctypes.c_void_p), ("trajectory", ctypes.c_void_p),
                    ("trajectory_len", ctypes.c_int), ("trajectory_row", ctypes.c_int),
                    ("subset_table_size", ctypes.c_int), ("subset_seed", ctypes.c_uint64), ("subset_buf", ctypes.c_void_p)]

    @torch.no_grad()
    def run_native(self, steps: int = 1, trajectory: Optional[torch.Tensor] = None, trajectory_row: int = 1) -> torch.Tensor:
        """``steps`` x ``step()`` as ONE call of ``mpx_rollout`` (fp32): same kernels, same order, bit-identical state --
        including the per-step scene re-render (``rerender_scene``) and the success tracking (``track_success``).
        ``trajectory`` [B, L, 7] (optional): rows ``trajectory_row ...`` of every environment receive q after each step."""
        if getattr(self, "_native", None) is None:
            c, sc = self.collision, self._NativeScene()
            sc.limits, sc.finger = self.limits.data_ptr(), float(self.sampler.finger)
            sc.table_pts, sc.table_link = self.sampler.table_pts.data_ptr(), self.sampler.table_link.data_ptr()
            sc.subset, sc.n_robot = self.subset.data_ptr(), int(self.subset.numel())
            sc.sph_centers, sc.sph_radii, sc.sph_link, sc.n_spheres = c.centers.data_ptr(), c.radii.data_ptr(), c.links.data_ptr(), c.num_spheres
            sc.cub_frames, sc.cub_dims, sc.M1 = self.cuboids.inv_frames.data_ptr(), self._cd.data_ptr(), self.cuboids.centers.size(1)
            sc.cyl_frames, sc.cyl_radii, sc.cyl_heights = self.cylinders.inv_frames.data_ptr(), self._yr.data_ptr(), self._yh.data_ptr()
            sc.M2 = self.cylinders.centers.size(1)
            w, keep = self.model.native_weights()
            need = _lib.load().mpx_rollout_workspace(self.B, self.xyz.size(1))
            self._native = (sc, w, keep, torch.empty(need, dtype=torch.uint8, device=self.device), need)
        sc, w, _, ws, need = self._native
        opt = self._NativeOptions()
        opt.steps, opt.first_step = int(steps), int(self.steps_done)
        keep = []
        if self.rerender_scene:
            prims = [_lib.f32c(self._prims[k]) for k in ("cuboid_centers", "cuboid_quats", "cylinder_centers", "cylinder_quats")]
            keep += prims
            opt.n_scene, opt.scene_seed, opt.env_offset = self._n_scene, self.scene_seed & (2 ** 64 - 1), self.env_offset
            opt.cub_centers, opt.cub_quats, opt.cyl_centers, opt.cyl_quats = (t.data_ptr() for t in prims)
        if self.done is not None:
            opt.target_poses, opt.pos_tol, opt.cos_rot_tol = self.targets.data_ptr(), self.pos_tol, self.cos_tol
            opt.done, opt.steps_taken = self.done.data_ptr(), self.steps.data_ptr()
        if self.resample_subset:
            opt.subset_table_size, opt.subset_seed = self.sampler.table_pts.size(0), self.subset_seed & (2 ** 64 - 1)
            opt.subset_buf = self.subset.data_ptr()
        if trajectory is not None:
            assert trajectory.is_contiguous() and trajectory.dtype == torch.float32 and trajectory.shape[::2] == (self.B, 7)
            opt.trajectory, opt.trajectory_len, opt.trajectory_row = trajectory.data_ptr(), trajectory.size(1), int(trajectory_row)
        _lib.call("mpx_rollout", ctypes.addressof(w), ctypes.addressof(sc), ctypes.addressof(opt), _lib.ptr(self.xyz),
                  self.xyz.size(1), _lib.ptr(self.q_norm), _lib.ptr(self.q), self.B, _lib.ptr(self.flags), None, _lib.ptr(ws),
                  need)
        del keep
        self.steps_done += int(steps)
        return self.q

    def step_native(self) -> torch.Tensor:
        """``step()`` through the single C entry point (one step of ``run_native``)."""
        return self.run_native(1)

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

    def rollout_until_success(self, max_steps: int = 150, check_every: int = 1, native: bool = False):
        """Batched ``rollout_until_success`` (run_inference.py:137-191): step until every environment is
        within 1 cm / 15 deg of its target or ``max_steps`` is reached.  The host looks at the done
        flags only every ``check_every`` steps (1 = the reference's per-step behaviour).  ``native``: every
        ``check_every`` steps are one ``mpx_rollout`` call (same result).

        :returns: trajectory [B, L+1, 7] and lengths int32 [B] (number of valid waypoints per env,
                  including the start configuration; later rows repeat the final configuration).
        """
        assert self.done is not None, "call track_success(target_poses) first"
        lim = self.limits
        if native:
            traj = torch.empty((self.B, max_steps + 1, 7), dtype=torch.float32, device=self.device)
            traj[:, 0] = (self.q_norm + 1) * (lim[:, 1] - lim[:, 0]) / 2 + lim[:, 0]
            taken = 0
            while taken < max_steps:
                n = min(check_every, max_steps - taken)
                self.run_native(n, trajectory=traj, trajectory_row=taken + 1)
                taken += n
                if bool(torch.all(self.done != 0)):
                    break
            return traj[:, :taken + 1], self.steps + 1
        traj = [(self.q_norm + 1) * (lim[:, 1] - lim[:, 0]) / 2 + lim[:, 0]]
        for i in range(max_steps):
            traj.append(self.step().clone())
            if (i + 1) % check_every == 0 and bool(torch.all(self.done != 0)):
                break
        return torch.stack(traj, dim=1), self.steps + 1


def rollout_until_success(mdl: MotionPolicyNetwork, q0, target, point_cloud: torch.Tensor, fk_sampler: FrankaSampler,
                          max_rollout_length: int = 150) -> np.ndarray:
    """Reference signature (run_inference.py:137-191) for one problem.

    :param q0: start configuration [7]; :param target: pose of ``right_gripper`` -- a 4x4 matrix or
        any object with ``.matrix``; :param point_cloud: [1, 2048+4096+128, 4] on the GPU (mutated in place).
    :rtype np.ndarray: the trajectory [T, 7], T <= max_rollout_length + 1
    """
    assert point_cloud.ndim == 3 and point_cloud.size(0) == 1
    dev = point_cloud.device
    q = torch.as_tensor(np.asarray(q0, dtype=np.float32)).reshape(1, 7).to(dev)
    lim = torch.as_tensor(ft.JOINT_LIMITS_REAL, dtype=torch.float32, device=dev)
    tm = torch.as_tensor(np.asarray(getattr(target, "matrix", target), dtype=np.float32)).reshape(1, 4, 4).to(dev)
    zeros = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
    prob = {"xyz": point_cloud, "q_norm": (q - lim[:, 0]) / (lim[:, 1] - lim[:, 0]) * 2 - 1,
            "cuboid_centers": zeros(1, 1, 3), "cuboid_dims": zeros(1, 1, 3),
            "cuboid_quats": torch.tensor([[[1.0, 0, 0, 0]]], device=dev), "cylinder_centers": zeros(1, 1, 3),
            "cylinder_radii": zeros(1, 1, 1), "cylinder_heights": zeros(1, 1, 1),
            "cylinder_quats": torch.tensor([[[1.0, 0, 0, 0]]], device=dev)}
    eng = RolloutEngine(mdl, prob, robot_subset=torch.zeros(2048, dtype=torch.int32, device=dev))
    eng.sampler = fk_sampler
    eng.track_success(tm)
    # The reference's loop (run_inference.py:171-189): forward, clamp, unnormalise, append, success test (one host
    # synchronisation per step, like its .cpu() there), and only when the test fails a fresh robot cloud -- whose column
    # subset robofin draws from np.random at every call.  The engine's step does the same work in one pass, so the
    # subset of step i is drawn in front of it and the draw is undone when the step turns out to be the last one: the
    # host RNG is left exactly where the reference leaves it.
    traj = [eng.q_norm.new_tensor(np.asarray(q0, dtype=np.float32)).reshape(1, 7)]
    for _ in range(max_rollout_length):
        state, rows = np.random.get_state(), point_cloud[:, :2048, :3].clone()
        eng.subset = fk_sampler.draw_subset(2048)
        traj.append(eng.step().clone())
        if bool(eng.done[0] != 0):  # (the reference breaks BEFORE it resamples: undo the draw and the cloud refresh)
            np.random.set_state(state)
            point_cloud[:, :2048, :3] = rows
            break
    return torch.cat(traj, dim=0).cpu().numpy()
