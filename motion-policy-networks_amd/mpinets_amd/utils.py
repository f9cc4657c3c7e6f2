"""Joint (un)normalisation with the reference's names and contracts (``mpinets/utils.py``).

``normalize_franka_joints`` / ``unnormalize_franka_joints`` accept ``torch.Tensor`` or
``np.ndarray`` of shape [7], [B,7] or [B,T,7] and raise ``NotImplementedError`` otherwise
(utils.py:127,244).  Arithmetic follows utils.py:91-93 and :207-209.  The range asserts of
``unnormalize`` (utils.py:200-201) force a device sync in the reference; here they are kept for
API parity but can be skipped with ``check=False`` on the hot path.
"""
from __future__ import annotations

from typing import Tuple, Union

import numpy as np
import torch

from .robot import FrankaRealRobot, FrankaRobot


def _check_dof(x, dof):
    assert (x.ndim == 1 and x.shape[0] == dof) or (x.ndim == 2 and x.shape[1] == dof) or (
        x.ndim == 3 and x.shape[2] == dof)


def normalize_franka_joints(batch_trajectory: Union[np.ndarray, torch.Tensor],
                            limits: Tuple[float, float] = (-1, 1),
                            use_real_constraints: bool = True):
    # the reference ignores `use_real_constraints` here and always uses the real limits
    # (utils.py:120,124); kept.
    robot = FrankaRealRobot
    if isinstance(batch_trajectory, torch.Tensor):
        lim = torch.as_tensor(robot.JOINT_LIMITS).type_as(batch_trajectory)
        _check_dof(batch_trajectory, robot.DOF)
        return (batch_trajectory - lim[:, 0]) / (lim[:, 1] - lim[:, 0]) * (limits[1] - limits[0]) + limits[0]
    elif isinstance(batch_trajectory, np.ndarray):
        lim = robot.JOINT_LIMITS
        _check_dof(batch_trajectory, robot.DOF)
        return (batch_trajectory - lim[:, 0]) / (lim[:, 1] - lim[:, 0]) * (limits[1] - limits[0]) + limits[0]
    raise NotImplementedError("Only torch.Tensor and np.ndarray implemented")


def unnormalize_franka_joints(batch_trajectory: Union[np.ndarray, torch.Tensor],
                              limits: Tuple[float, float] = (-1, 1),
                              use_real_constraints: bool = True, check: bool = True):
    robot = FrankaRealRobot if use_real_constraints else FrankaRobot
    if isinstance(batch_trajectory, torch.Tensor):
        lim = torch.as_tensor(robot.JOINT_LIMITS).type_as(batch_trajectory)
        _check_dof(batch_trajectory, lim.size(0))
        if check:
            assert torch.all(batch_trajectory >= limits[0])
            assert torch.all(batch_trajectory <= limits[1])
        rng, lo = lim[:, 1] - lim[:, 0], lim[:, 0]
        return (batch_trajectory - limits[0]) * rng / (limits[1] - limits[0]) + lo
    elif isinstance(batch_trajectory, np.ndarray):
        lim = robot.JOINT_LIMITS
        _check_dof(batch_trajectory, robot.DOF)
        if check:
            assert np.all(batch_trajectory >= limits[0])
            assert np.all(batch_trajectory <= limits[1])
        return (batch_trajectory - limits[0]) * (lim[:, 1] - lim[:, 0]) / (limits[1] - limits[0]) + lim[:, 0]
    raise NotImplementedError("Only torch.Tensor and np.ndarray implemented")
