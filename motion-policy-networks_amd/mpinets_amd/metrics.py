"""Batched, PyBullet-free trajectory evaluation on the device (next row N3 of SURVEY.md section 8f).

The reference's ``Evaluator.evaluate_trajectory`` (``mpinets/metrics.py:436-523``) scores ONE
trajectory at a time with PyBullet (and optionally Lula) collision checkers on the host.  This
module scores a whole batch ``[B, T, 7]`` on the GPU with the metrics that need no physics engine,
under the reference's names:

=================================  ==============================================================
``position_error`` [cm]            ``check_final_position`` (metrics.py:338-347)
``orientation_error`` [deg]        ``check_final_orientation`` (metrics.py:349-361)
``eff_position_path_length`` [m]   ``calculate_eff_path_lengths`` (metrics.py:410-434)
``eff_orientation_path_length``    same, degrees
``joint_limit_violation``          ``violates_joint_limits`` (metrics.py:311-322), published limits
``collision``                      swept-sphere SDF check of ``model.py:293-314`` -- NOT PyBullet/Lula
``self_collision``                 body-cylinder vs end-link spheres of ``config/franka_fabric_config.yaml``
                                   -- NOT PyBullet
``correct_final_region``           ``check_final_region`` (metrics.py:363-384) with SDF volumes
``success``                        position < 1 cm, orientation < 15 deg, region ok, no violation
                                   (metrics.py:514-519)
``config_smoothness``              ``calculate_smoothness`` (metrics.py:387-409): SPARC of the joint-space and of the
``eff_smoothness``                 end-effector speed profile (``smoothness.py``: one batched FFT per FFT length;
                                   pinned to the reference's own ``third_party/sparc.py``), when ``dt`` is given
=================================  ==============================================================

``BatchedEvaluator.metrics`` folds such a result into the summary of ``Evaluator.metrics`` (metrics.py:566-664) under
the reference's keys -- those that need neither wall-clock times nor PyBullet penetration depths.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import _lib
from . import franka_tables as ft
from .robot import FrankaCollisionSampler


def _per_primitive_sdf(volumes, points: torch.Tensor) -> torch.Tensor:
    """SDF of EVERY primitive of a set at one point per environment: points [B,3] -> [B,M] (zero-volume rows: +inf).
    The set classes reduce over their primitives; viewing each primitive as its own one-primitive environment gives
    the unreduced values from the same kernels."""
    from .geometry import TorchCuboids, TorchCylinders, TorchSpheres

    B, M = volumes.centers.shape[:2]
    flat = lambda t: t.reshape(B * M, 1, t.size(-1))
    if isinstance(volumes, TorchCuboids):
        one = TorchCuboids(flat(volumes.centers), flat(volumes.dims), flat(volumes.quats))
    elif isinstance(volumes, TorchCylinders):
        one = TorchCylinders(flat(volumes.centers), flat(volumes.radii), flat(volumes.heights), flat(volumes.quats))
    elif isinstance(volumes, TorchSpheres):
        one = TorchSpheres(flat(volumes.centers), flat(volumes.radii))
    else:
        raise TypeError(f"unsupported volume set {type(volumes).__name__}")
    p = points[:, None, :].expand(B, M, 3).reshape(B * M, 1, 3).contiguous()
    return one.sdf(p).reshape(B, M)


class BatchedEvaluator:
    def __init__(self, device, finger: float = ft.FINGER_OPENING):
        self.device = torch.device(device)
        self.finger = float(finger)
        self.limits = torch.as_tensor(ft.JOINT_LIMITS_PUBLISHED, dtype=torch.float32, device=self.device).contiguous()
        self.collision_sampler = FrankaCollisionSampler(self.device, with_base_link=False, finger=finger)

    @torch.no_grad()
    def evaluate_trajectories(self, trajectories: torch.Tensor, target_poses: torch.Tensor,
                              lengths: Optional[torch.Tensor] = None, cuboids=None, cylinders=None,
                              target_volume=None, negative_volumes=None, dt: Optional[float] = None) -> Dict[str, torch.Tensor]:
        """:param trajectories: [B,T,7] joint angles (rows past ``lengths[b]`` are ignored; they must still be
            valid configurations, e.g. the final one repeated, for the collision sweep)
        :param target_poses: [B,4,4] ``right_gripper`` targets
        :param cuboids/cylinders: scene primitives (``geometry.TorchCuboids`` / ``TorchCylinders``) or None
        :param target_volume: optional primitive set (M >= 1 per env); the final position must be inside one
        :param negative_volumes: optional primitive set; the final position must be outside all of them -- except
            those that contain the TARGET position, which the reference drops first ("Sometimes the target is inside a
            negative volume. This is obviously a bad negative volume", metrics.py:507-512)
        :param dt: time between waypoints; when given (and T >= 2) the result also carries ``config_smoothness`` and
            ``eff_smoothness`` (SPARC, metrics.py:387-409, 495-497; the reference evaluates at dt = 0.12 s)
        """
        _lib.require_cuda(trajectories, target_poses)
        B, T, _ = trajectories.shape
        dev = trajectories.device
        tr, tg = _lib.f32c(trajectories), _lib.f32c(target_poses)
        ln = None if lengths is None else _lib.i32c(lengths)
        f = lambda: torch.empty(B, dtype=torch.float32, device=dev)
        i = lambda: torch.zeros(B, dtype=torch.int32, device=dev)
        pos, ori, pp, po, jl, sc = f(), f(), f(), f(), i(), i()
        _lib.call("mpx_trajectory_metrics", _lib.ptr(tr), _lib.ptr(ln), _lib.ptr(tg), _lib.ptr(self.limits), B, T,
                  self.finger, _lib.ptr(pos), _lib.ptr(ori), _lib.ptr(pp), _lib.ptr(po), _lib.ptr(jl), _lib.ptr(sc))
        if ln is not None:  # freeze the tail so the swept-sphere check only sees valid waypoints
            t_idx = torch.minimum(torch.arange(T, device=dev)[None, :], (ln.long() - 1).clamp(min=0)[:, None])
            tr = torch.gather(tr, 1, t_idx[:, :, None].expand(-1, -1, 7)).contiguous()
        collision = self.collision_sampler.check(tr, cuboids, cylinders)
        region = torch.ones(B, dtype=torch.bool, device=dev)
        if target_volume is not None or negative_volumes is not None:
            last = (ln.long() - 1).clamp(min=0) if ln is not None else torch.full((B,), T - 1, device=dev)
            from .robot import franka_fk

            final = franka_fk(tr[torch.arange(B, device=dev), last], self.finger)[:, ft.LINK_ID["right_gripper"], 9:]
            p = final[:, None, :].contiguous()
            if target_volume is not None:
                region &= target_volume.sdf(p)[:, 0] <= 0
            if negative_volumes is not None:
                # corrected_negative_volumes = [v for v in volumes if v.sdf(target) > 0]  (metrics.py:507-509)
                keep = _per_primitive_sdf(negative_volumes, tg[:, :3, 3].contiguous()) > 0  # [B,M]; masked rows: +inf
                region &= ~((_per_primitive_sdf(negative_volumes, final.contiguous()) <= 0) & keep).any(dim=1)
        jl, sc = jl != 0, sc != 0
        violation = collision | jl | sc
        res = {"position_error": pos, "orientation_error": ori, "eff_position_path_length": pp,
               "eff_orientation_path_length": po, "joint_limit_violation": jl, "self_collision": sc,
               "collision": collision, "physical_violations": violation, "correct_final_region": region,
               "success": (pos < 1) & (ori < 15) & region & ~violation,
               "num_steps": ln if ln is not None else torch.full((B,), T, dtype=torch.int32, device=dev)}
        if dt is not None and T >= 2:
            res["config_smoothness"], res["eff_smoothness"] = self.smoothness(tr, ln, dt)
        return res

    @torch.no_grad()
    def smoothness(self, trajectories: torch.Tensor, lengths: Optional[torch.Tensor], dt: float):
        """SPARC of the joint-space and of the ``right_gripper`` speed profile of every trajectory [B,T,7]
        (``calculate_smoothness``, metrics.py:387-409) -> (config_sparc [B], eff_sparc [B]) float64."""
        from .robot import franka_fk
        from .smoothness import trajectory_smoothness

        B, T, _ = trajectories.shape
        tr = _lib.f32c(trajectories)
        eff = franka_fk(tr.reshape(B * T, 7), self.finger)[:, ft.LINK_ID["right_gripper"], 9:].reshape(B, T, 3)
        return trajectory_smoothness(tr, eff, lengths, dt)

    @staticmethod
    def metrics(results: Dict[str, torch.Tensor]) -> Dict[str, object]:
        """``Evaluator.metrics`` (metrics.py:566-664) over a result of ``evaluate_trajectories``: percentages and means
        under the reference's keys.  Not reproduced: ``time`` / ``step time`` (the caller's clock), the collision depths
        (PyBullet penetration queries) and ``skips`` (the reference's planner-failure bookkeeping)."""
        pct = lambda t: 100.0 * float(t.double().mean().item())  # percent_true (metrics.py:50-62)
        ok = results["success"]
        out = {"success": pct(ok), "total": int(ok.numel()),
               "env collision": pct(results["collision"]), "self collision": pct(results["self_collision"]),
               "joint violation": pct(results["joint_limit_violation"]),
               "physical violations": pct(results["physical_violations"]),
               "1 cm": pct(results["position_error"] < 1), "5 cm": pct(results["position_error"] < 5),
               "15 deg": pct(results["orientation_error"] < 15), "30 deg": pct(results["orientation_error"] < 30),
               "165 deg": pct(results["orientation_error"] > 165)}
        for key, name in (("eff_position_path_length", "eff position path length"),
                          ("eff_orientation_path_length", "eff orientation path length")):
            v = results[key][ok].double()  # (successful trajectories only, metrics.py:609-622; numpy's population std)
            out[name] = (float(v.mean().item()), float(v.std(unbiased=False).item())) if v.numel() else (float("nan"), float("nan"))
        if "config_smoothness" in results:
            cs, es = results["config_smoothness"], results["eff_smoothness"]
            out["is smooth"] = pct((cs < -1.6) & (es < -1.6))
            out["average config sparc"], out["average eff sparc"] = float(cs.mean().item()), float(es.mean().item())
        return out
