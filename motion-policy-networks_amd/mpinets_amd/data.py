"""Batch assembly on the device from the dataset arrays (next row N2 of SURVEY.md section 8f).

The reference keeps the dataset in one HDF5 file per split and builds each sample in a CPU DataLoader worker
(``mpinets/data_loader.py``: ``PointCloudBase.get_inputs`` :141-280, ``PointCloudTrajectoryDataset`` :283-342,
``PointCloudInstanceDataset`` :345-417).  Here the arrays of that schema are uploaded once (the full training set
is ~10 GB of the 288 GB of HBM) and a batch is assembled by kernels: joint noise + clamp + normalise + target FK
(``mpx_batch_configs``), primitive rows gathered by trajectory (``mpx_gather_rows``), robot / target / scene
clouds written straight into the ``[B, 6272, 4]`` slab (``mpx_franka_cloud``, ``mpx_pose_cloud``,
``mpx_scene_cloud``).  Class and key names follow the reference; ``get_batch`` is the engine's native call,
``__getitem__`` gives the reference's per-sample dict.

HDF5 schema (``data_pipeline/gen_data.py``): ``cuboid_{dims,centers,quaternions}`` [N,Mc,3|3|4],
``cylinder_{radii,heights,centers,quaternions}`` [N,My,1|1|3|4] (optional), ``{hybrid,global}_solutions``
[N,50,7].  Sources accepted: a mapping of arrays with those keys, an ``.npz`` with those keys, an ``.hdf5`` file or
the reference's directory layout (``<dir>/{train,val,test}/**/*.hdf5``) -- the last two need ``h5py``, which this
image does not ship (the loader raises a clear error instead of guessing).
"""
from __future__ import annotations

import enum
import os
from pathlib import Path
from typing import Dict, Iterator, Mapping, Optional, Sequence, Union

import numpy as np
import torch

from . import _lib
from . import franka_tables as ft
from .robot import FrankaSampler
from .scenes import sample_scene_clouds

PRIM_KEYS = {"cuboid_dims": "cuboid_dims", "cuboid_centers": "cuboid_centers", "cuboid_quaternions": "cuboid_quats",
             "cylinder_radii": "cylinder_radii", "cylinder_heights": "cylinder_heights",
             "cylinder_centers": "cylinder_centers", "cylinder_quaternions": "cylinder_quats"}


class DatasetType(enum.IntEnum):  # data_loader.py:37-45
    TRAIN = 0
    VAL = 1
    TEST = 2


def _load_arrays(source, dataset_type: DatasetType) -> Mapping[str, np.ndarray]:
    if isinstance(source, Mapping):
        return source
    path = Path(source)
    if path.is_dir():  # data_loader.py:103-122
        sub = path / {DatasetType.TRAIN: "train", DatasetType.VAL: "val", DatasetType.TEST: "test"}[dataset_type]
        found = list(sub.glob("**/*.hdf5")) + list(sub.glob("**/*.npz"))
        assert len(found) == 1, f"expected exactly one database under {sub}, found {len(found)}"
        path = found[0]
    if path.suffix == ".npz":
        return dict(np.load(path))
    try:
        import h5py
    except ImportError as e:  # pragma: no cover - h5py is absent from this image
        raise _lib.MpxError(f"reading {path} needs h5py, which is not installed; convert the file to .npz with the "
                            "same keys or pass the arrays directly") from e
    with h5py.File(str(path), "r") as f:
        return {k: f[k][...] for k in f.keys()}


class PointCloudBase:
    """Dataset resident on the GPU + batched ``get_inputs`` (data_loader.py:48-280)."""

    def __init__(self, directory: Union[str, os.PathLike, Mapping[str, np.ndarray]], trajectory_key: str,
                 num_robot_points: int, num_obstacle_points: int, num_target_points: int,
                 dataset_type: DatasetType, random_scale: float, device="cuda:0", seed: int = 0):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.MpxError("the batch assembler needs a GPU device (no CPU fallback)")
        self.type = DatasetType(dataset_type)
        self.trajectory_key = trajectory_key
        self.train = self.type == DatasetType.TRAIN
        self.num_robot_points, self.num_obstacle_points = num_robot_points, num_obstacle_points
        self.num_target_points = num_target_points
        self.random_scale = float(random_scale)
        self.seed = int(seed)
        arrays = _load_arrays(directory, self.type)
        up = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).to(self.device)
        self.trajectories = up(arrays[trajectory_key])
        self._num_trajectories, self.expert_length = int(self.trajectories.size(0)), int(self.trajectories.size(1))
        n = self._num_trajectories
        self.prims: Dict[str, torch.Tensor] = {}
        for src, dst in PRIM_KEYS.items():
            if src in arrays:
                a = np.asarray(arrays[src], dtype=np.float32)
                if a.ndim == 2:  # a single primitive per scene is stored without the M axis (data_loader.py:192-201)
                    a = a[:, None, :]
                self.prims[dst] = up(a)
        if "cylinder_radii" not in self.prims:  # dummy cylinder (data_loader.py:213-218)
            z = lambda *s: torch.zeros((n,) + s, dtype=torch.float32, device=self.device)
            self.prims.update(cylinder_radii=z(1, 1), cylinder_heights=z(1, 1), cylinder_centers=z(1, 3),
                              cylinder_quats=z(1, 4))
        for k in ("cuboid_quats", "cylinder_quats"):  # all-zero quaternions -> unit (data_loader.py:203-208, :232)
            qz = self.prims[k]
            bad = (qz.abs() <= 1e-8).all(dim=-1)
            qz[..., 0] = torch.where(bad, torch.ones_like(qz[..., 0]), qz[..., 0])
        self.limits = torch.as_tensor(ft.JOINT_LIMITS_REAL, dtype=torch.float32, device=self.device).contiguous()
        self.fk_sampler = FrankaSampler(self.device, use_cache=True)
        self._calls = 0

    @property
    def num_trajectories(self) -> int:
        return self._num_trajectories

    @torch.no_grad()
    def get_inputs_batch(self, trajectory_idx: torch.Tensor, timestep: Optional[torch.Tensor],
                         with_supervision: bool, seed: Optional[int] = None, sample_offset: int = 0,
                         robot_subset: Optional[torch.Tensor] = None,
                         target_subset: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """Batched ``get_inputs`` (+ the supervision row of ``PointCloudInstanceDataset.__getitem__``).  Every random
        draw of row b is keyed by (seed, ``sample_offset + b``): a rank holding rows [o, o+n) of a global batch passes
        ``sample_offset=o`` and draws what a single process would.  ``robot_subset`` / ``target_subset``: point-table
        rows to use for the robot / target clouds instead of the two host draws (tests replay the reference's)."""
        dev = self.device
        ti = torch.as_tensor(trajectory_idx, dtype=torch.int64, device=dev).contiguous()
        ts = None if timestep is None else torch.as_tensor(timestep, dtype=torch.int32, device=dev).contiguous()
        B = int(ti.numel())
        seed = self.seed + 0x9E3779B1 * self._calls if seed is None else int(seed)
        self._calls += 1
        f = lambda *s: torch.empty((B,) + s, dtype=torch.float32, device=dev)
        q, qn, pose, pos = f(7), f(7), f(4, 4), f(3)
        sup = f(7) if with_supervision else None
        _lib.call("mpx_batch_configs", _lib.ptr(self.trajectories), self._num_trajectories, self.expert_length,
                  _lib.ptr(ti), _lib.ptr(ts), _lib.ptr(self.limits), self.random_scale if self.train else 0.0,
                  seed & (2 ** 64 - 1), int(sample_offset), int(self.train), B, self.fk_sampler.finger, _lib.ptr(q), _lib.ptr(qn), _lib.ptr(sup),
                  _lib.ptr(pose), _lib.ptr(pos))
        item = {"configuration": qn, "target_position": pos}
        if sup is not None:
            item["supervision"] = sup
        for k, src in self.prims.items():
            row = int(np.prod(src.shape[1:]))
            dst = torch.empty((B,) + tuple(src.shape[1:]), dtype=torch.float32, device=dev)
            _lib.call("mpx_gather_rows", _lib.ptr(src), _lib.ptr(ti), B, row, _lib.ptr(dst))
            item[k] = dst
        nr, no, nt = self.num_robot_points, self.num_obstacle_points, self.num_target_points
        xyz = torch.empty((B, nr + no + nt, 4), dtype=torch.float32, device=dev)
        xyz[:, :nr, 3] = 0  # label column (data_loader.py:261-267)
        xyz[:, nr:nr + no, 3] = 1
        xyz[:, nr + no:, 3] = 2
        if robot_subset is None:
            robot_subset = self.fk_sampler.draw_subset(nr)
        robot_subset = robot_subset.to(device=dev, dtype=torch.int32).contiguous()
        assert robot_subset.numel() == nr
        self.fk_sampler.sample_into(q, xyz, robot_subset)
        sample_scene_clouds(item, no, seed ^ 0x5CE7E, out=xyz[:, nr:nr + no], env_offset=sample_offset)
        tgt = self.fk_sampler.sample_end_effector(pose, num_points=nt, subset=target_subset)
        xyz[:, nr + no:, :3] = tgt
        item["xyz"] = xyz
        return item

    def _single(self, item: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        return {k: v[0] for k, v in item.items()}


class PointCloudTrajectoryDataset(PointCloudBase):
    """One element = one trajectory start + scene, no supervision (validation; data_loader.py:283-342)."""

    def __init__(self, directory, trajectory_key: str, num_robot_points: int, num_obstacle_points: int,
                 num_target_points: int, dataset_type: DatasetType, device="cuda:0", seed: int = 0):
        assert dataset_type != DatasetType.TRAIN, "This dataset is not meant for training"
        super().__init__(directory, trajectory_key, num_robot_points, num_obstacle_points, num_target_points,
                         dataset_type, random_scale=0.0, device=device, seed=seed)

    def __len__(self) -> int:
        return self.num_trajectories

    def get_batch(self, indices: Sequence[int], seed: Optional[int] = None) -> Dict[str, torch.Tensor]:
        return self.get_inputs_batch(torch.as_tensor(indices), None, with_supervision=False, seed=seed)

    def __getitem__(self, idx: int) -> Dict[str, torch.Tensor]:
        return self._single(self.get_batch([idx]))


class PointCloudInstanceDataset(PointCloudBase):
    """One element = one waypoint of one trajectory, supervised by the next waypoint (data_loader.py:345-417)."""

    def __len__(self) -> int:
        return self.num_trajectories * self.expert_length

    def get_batch(self, indices: Sequence[int], seed: Optional[int] = None) -> Dict[str, torch.Tensor]:
        idx = torch.as_tensor(indices, dtype=torch.int64)
        ti = torch.div(idx, self.expert_length, rounding_mode="floor")
        ts = (idx - ti * self.expert_length).to(torch.int32)
        return self.get_inputs_batch(ti, ts, with_supervision=True, seed=seed)

    def __getitem__(self, idx: int) -> Dict[str, torch.Tensor]:
        return self._single(self.get_batch([idx]))

    def batches(self, batch_size: int, shuffle: bool = True, seed: int = 0, rank: int = 0, world_size: int = 1,
                drop_last: bool = True) -> Iterator[Dict[str, torch.Tensor]]:
        """One epoch of device-resident batches; ranks take disjoint strided shares of the (shuffled) index list."""
        n = len(self)
        order = np.random.default_rng(seed).permutation(n) if shuffle else np.arange(n)
        order = order[rank::world_size]
        stop = len(order) - (len(order) % batch_size if drop_last else 0)
        for i in range(0, stop, batch_size):
            yield self.get_batch(order[i:i + batch_size])
