"""Franka Panda data tables shipped by the engine (data, not algorithm).

The reference takes all of these from the un-vendored ``robofin`` v0.0.1
(``/root/reference/docker/Dockerfile:153``; call sites ``mpinets/model.py:250,267-271,300``,
``mpinets/utils.py:50-51``).  None of it is in the reference tree, so this file is the
engine's own, explicit, diff-able statement of the robot:

* ``JOINT_LIMITS_REAL`` / ``JOINT_LIMITS_PUBLISHED`` -- ``FrankaRealRobot.JOINT_LIMITS`` /
  ``FrankaRobot.JOINT_LIMITS`` stand-ins ([EXT-RECALL]).
* ``LINK_NAMES`` -- frame ids produced by the FK kernel (csrc/franka.hip).
* ``COLLISION_SPHERES`` -- the in-repo sphere model,
  ``/root/reference/config/franka_robot_description.yaml:57-182`` (57 spheres, 10 radii).
* ``link_point_table()`` -- robot-surface point table.  robofin samples link *meshes*; no mesh
  ships with the reference or this image, so the table is sampled on the surface of the union
  of each link's collision spheres (deterministic Fibonacci lattices).  Users who have the
  meshes can pass their own ``(points, link_ids)`` to ``FrankaSampler(point_table=...)``.

Kinematic constants are the public Franka Panda URDF values (d1 .333, d3 .316, a4 .0825,
a5 -.0825, d5 .384, a7 .088, flange .107, hand yaw -pi/4, finger z .0584).
"""
from __future__ import annotations

import functools
from typing import Dict, List, Tuple

import numpy as np

DOF = 7

# [EXT-RECALL] robofin FrankaRealRobot.JOINT_LIMITS (empirical limits used by the reference's
# normalisation, mpinets/utils.py:84-93).
JOINT_LIMITS_REAL = np.array(
    [
        (-2.3093, 2.3093),
        (-1.5133, 1.5133),
        (-2.4937, 2.4937),
        (-2.7478, -0.4461),
        (-2.4800, 2.4800),
        (0.8521, 4.2094),
        (-2.6895, 2.6895),
    ],
    dtype=np.float64,
)

# Published Franka limits (robofin FrankaRobot.JOINT_LIMITS stand-in).
JOINT_LIMITS_PUBLISHED = np.array(
    [
        (-2.8973, 2.8973),
        (-1.7628, 1.7628),
        (-2.8973, 2.8973),
        (-3.0718, -0.0698),
        (-2.8973, 2.8973),
        (-0.0175, 3.7525),
        (-2.8973, 2.8973),
    ],
    dtype=np.float64,
)

# /root/reference/config/franka_robot_description.yaml:44-46
DEFAULT_Q = np.array([0.00, -1.3, 0.00, -2.87, 0.00, 2.00, 0.75], dtype=np.float64)
# /root/reference/config/franka_robot_description.yaml:51-53
FINGER_OPENING = 0.025

# /root/reference/config/franka_fabric_config.yaml:117-140: the Geometric-Fabrics self-collision model the batched
# metrics use (csrc/franka.hip trajectory_metrics_kernel): the base body cylinder as a capped segment + radius, and the
# spheres tested against it.  Of the fabric's seven named spheres the kernel uses the four that sit on frames the FK
# produces (link7, hand, the two fingertips); panda_wrist_end_pt / panda_face_{left,right} are fabric-only task frames.
FABRIC_BODY_CYLINDER = {"pt1": (0.0, 0.0, 0.333), "pt2": (0.0, 0.0, -0.3), "radius": 0.15}
FABRIC_SELF_SPHERES = (("panda_link7", 0.1), ("panda_hand", 0.01), ("panda_leftfingertip", 0.01),
                       ("panda_rightfingertip", 0.01))

# Frame ids written by the FK kernel, in this order (15 frames x 3x4 row-major floats).
LINK_NAMES: Tuple[str, ...] = (
    "panda_link0",
    "panda_link1",
    "panda_link2",
    "panda_link3",
    "panda_link4",
    "panda_link5",
    "panda_link6",
    "panda_link7",
    "panda_link8",
    "panda_hand",
    "panda_leftfinger",
    "panda_rightfinger",
    "panda_leftfingertip",
    "panda_rightfingertip",
    "right_gripper",
)
LINK_ID: Dict[str, int] = {n: i for i, n in enumerate(LINK_NAMES)}
NUM_FRAMES = len(LINK_NAMES)

# /root/reference/config/franka_robot_description.yaml:57-182  (link -> [(center, radius)])
COLLISION_SPHERES: Dict[str, List[Tuple[Tuple[float, float, float], float]]] = {
    "panda_link0": [((0.0, 0.0, 0.05), 0.08)],
    "panda_link1": [
        ((0.0, -0.08, 0.0), 0.06),
        ((0.0, -0.03, 0.0), 0.06),
        ((0.0, 0.0, -0.12), 0.06),
        ((0.0, 0.0, -0.17), 0.06),
    ],
    "panda_link2": [
        ((0.0, 0.0, 0.03), 0.06),
        ((0.0, 0.0, 0.08), 0.06),
        ((0.0, -0.12, 0.0), 0.06),
        ((0.0, -0.17, 0.0), 0.06),
    ],
    "panda_link3": [
        ((0.0, 0.0, -0.06), 0.05),
        ((0.0, 0.0, -0.1), 0.06),
        ((0.08, 0.06, 0.0), 0.055),
        ((0.08, 0.02, 0.0), 0.055),
    ],
    "panda_link4": [
        ((0.0, 0.0, 0.02), 0.055),
        ((0.0, 0.0, 0.06), 0.055),
        ((-0.08, 0.095, 0.0), 0.06),
        ((-0.08, 0.06, 0.0), 0.055),
    ],
    "panda_link5": [
        ((0.0, 0.055, 0.0), 0.06),
        ((0.0, 0.075, 0.0), 0.06),
        ((0.0, 0.000, -0.22), 0.06),
        ((0.0, 0.05, -0.18), 0.05),
        ((0.01, 0.08, -0.14), 0.025),
        ((0.01, 0.085, -0.11), 0.025),
        ((0.01, 0.09, -0.08), 0.025),
        ((0.01, 0.095, -0.05), 0.025),
        ((-0.01, 0.08, -0.14), 0.025),
        ((-0.01, 0.085, -0.11), 0.025),
        ((-0.01, 0.09, -0.08), 0.025),
        ((-0.01, 0.095, -0.05), 0.025),
    ],
    "panda_link6": [
        ((0.0, 0.0, 0.0), 0.06),
        ((0.08, 0.03, 0.0), 0.06),
        ((0.08, -0.01, 0.0), 0.06),
    ],
    "panda_link7": [
        ((0.0, 0.0, 0.07), 0.05),
        ((0.02, 0.04, 0.08), 0.025),
        ((0.04, 0.02, 0.08), 0.025),
        ((0.04, 0.06, 0.085), 0.02),
        ((0.06, 0.04, 0.085), 0.02),
    ],
    "panda_hand": [
        ((0.0, -0.075, 0.01), 0.028),
        ((0.0, -0.045, 0.01), 0.028),
        ((0.0, -0.015, 0.01), 0.028),
        ((0.0, 0.015, 0.01), 0.028),
        ((0.0, 0.045, 0.01), 0.028),
        ((0.0, 0.075, 0.01), 0.028),
        ((0.0, -0.075, 0.03), 0.026),
        ((0.0, -0.045, 0.03), 0.026),
        ((0.0, -0.015, 0.03), 0.026),
        ((0.0, 0.015, 0.03), 0.026),
        ((0.0, 0.045, 0.03), 0.026),
        ((0.0, 0.075, 0.03), 0.026),
        ((0.0, -0.075, 0.05), 0.024),
        ((0.0, -0.045, 0.05), 0.024),
        ((0.0, -0.015, 0.05), 0.024),
        ((0.0, 0.015, 0.05), 0.024),
        ((0.0, 0.045, 0.05), 0.024),
        ((0.0, 0.075, 0.05), 0.024),
    ],
    "panda_leftfingertip": [((0.0, 0.0075, 0.0), 0.0108)],
    "panda_rightfingertip": [((0.0, -0.0075, 0.0), 0.0108)],
}


def collision_sphere_table(with_base_link: bool = False):
    """Flat sphere table, grouped by radius in order of first appearance.

    Mirrors what ``FrankaCollisionSampler.compute_spheres`` iterates over
    (reference call site ``mpinets/model.py:300-312``: a list of ``(radius, spheres)``).

    :returns: ``centers float32[S,3]``, ``radii float32[S]``, ``link_ids int32[S]``,
              ``groups = [(radius, start, count), ...]`` (contiguous ranges into the table)
    """
    flat = []
    for link, spheres in COLLISION_SPHERES.items():
        if link == "panda_link0" and not with_base_link:
            continue
        for c, r in spheres:
            flat.append((r, LINK_ID[link], c))
    radii_order: List[float] = []
    for r, _, _ in flat:
        if r not in radii_order:
            radii_order.append(r)
    centers, radii, links, groups = [], [], [], []
    for r in radii_order:
        start = len(centers)
        for rr, lid, c in flat:
            if rr == r:
                centers.append(c)
                radii.append(rr)
                links.append(lid)
        groups.append((float(r), start, len(centers) - start))
    return (
        np.asarray(centers, dtype=np.float32),
        np.asarray(radii, dtype=np.float32),
        np.asarray(links, dtype=np.int32),
        groups,
    )


def _fibonacci_sphere(n: int) -> np.ndarray:
    """n quasi-uniform unit vectors (golden-angle lattice), float64."""
    i = np.arange(n, dtype=np.float64) + 0.5
    z = 1.0 - 2.0 * i / n
    r = np.sqrt(np.maximum(0.0, 1.0 - z * z))
    phi = i * (np.pi * (3.0 - np.sqrt(5.0)))
    return np.stack([r * np.cos(phi), r * np.sin(phi), z], axis=1)


@functools.lru_cache(maxsize=4)
def link_point_table(total_points: int = 4096, with_base_link: bool = True):
    """Deterministic robot-surface point table in link-local coordinates.

    Stand-in for robofin's cached per-link mesh samples (SURVEY.md section 8 row a8,
    [EXT-RECALL]: ~4096 points allotted by link area, one table shared by the batch).
    Points lie on the boundary of the union of each link's collision spheres; the per-link
    allotment is proportional to that exposed area.

    :returns: ``points float32[P,3]`` (link-local), ``link_ids int32[P]``; P == total_points
    """
    dense = 4000
    per_link = []
    for link, spheres in COLLISION_SPHERES.items():
        if link == "panda_link0" and not with_base_link:
            continue
        cs = np.asarray([c for c, _ in spheres], dtype=np.float64)
        rs = np.asarray([r for _, r in spheres], dtype=np.float64)
        pts, wts = [], []
        for k in range(len(rs)):
            p = _fibonacci_sphere(dense) * rs[k] + cs[k]
            keep = np.ones(dense, dtype=bool)
            for j in range(len(rs)):
                if j != k:
                    keep &= np.linalg.norm(p - cs[j], axis=1) >= rs[j] - 1e-12
            pts.append(p[keep])
            # each lattice point stands for an equal share of its sphere's area
            wts.append(np.full(int(keep.sum()), 4.0 * np.pi * rs[k] ** 2 / dense))
        per_link.append((LINK_ID[link], np.concatenate(pts), np.concatenate(wts)))
    areas = np.array([w.sum() for _, _, w in per_link])
    alloc = np.floor(areas / areas.sum() * total_points).astype(int)
    # hand out the remainder to the largest fractional parts (deterministic)
    rem = total_points - alloc.sum()
    frac = areas / areas.sum() * total_points - alloc
    for i in np.argsort(-frac, kind="stable")[:rem]:
        alloc[i] += 1
    out_p, out_l = [], []
    for (lid, p, w), n in zip(per_link, alloc):
        # area-stratified pick: walk the cumulative area at n evenly spaced quantiles
        cum = np.cumsum(w)
        q = (np.arange(n) + 0.5) / n * cum[-1]
        idx = np.minimum(np.searchsorted(cum, q), len(p) - 1)
        out_p.append(p[idx])
        out_l.append(np.full(n, lid, dtype=np.int32))
    return (
        np.ascontiguousarray(np.concatenate(out_p), dtype=np.float32),
        np.ascontiguousarray(np.concatenate(out_l), dtype=np.int32),
    )


def load_point_tables(path: str):
    """Tables written by ``tools/dump_robofin_tables.py`` (run where robofin is installed) -> the arguments the
    samplers accept: ``{"point_table": (points [P,3], link_ids [P]), "joint_limits_real": [7,2], ...}``.
    ``FrankaSampler(device, point_table=load_point_tables(p)["point_table"])`` then samples robofin's own mesh points."""
    d = np.load(path, allow_pickle=False)
    names = [str(n) for n in d["point_link_name"]]
    unknown = sorted(set(names) - set(LINK_ID))
    if unknown:
        raise ValueError(f"{path}: links {unknown} are not frames of the FK kernel ({LINK_NAMES})")
    out = {"point_table": (np.ascontiguousarray(d["points"], dtype=np.float32),
                           np.asarray([LINK_ID[n] for n in names], dtype=np.int32))}
    for k in ("joint_limits_real", "joint_limits_published"):
        if k in d.files:
            out[k] = np.asarray(d[k], dtype=np.float64)
    return out


@functools.lru_cache(maxsize=2)
def end_effector_point_table(total_points: int = 512, frame: str = "right_gripper"):
    """Gripper point table expressed in ``frame`` (default ``right_gripper``).

    Stand-in for robofin's ``sample_end_effector`` table: the
    hand + finger(tip) points of ``link_point_table`` re-expressed in the end-effector frame
    at finger opening ``FINGER_OPENING``.
    """
    assert frame in ("right_gripper", "panda_link8", "panda_hand")
    pts, lids = link_point_table()
    sel = np.isin(lids, [LINK_ID[n] for n in ("panda_hand", "panda_leftfingertip", "panda_rightfingertip")])
    p = pts[sel].astype(np.float64)
    l = lids[sel]
    s = np.sqrt(0.5)
    # link8 -> hand: Rz(-pi/4); hand -> finger: (0, +-0.025, 0.0584); finger -> tip: (0,0,0.045)
    R_hand = np.array([[s, s, 0.0], [-s, s, 0.0], [0.0, 0.0, 1.0]])
    in_hand = p.copy()
    in_hand[l == LINK_ID["panda_leftfingertip"]] += np.array([0.0, FINGER_OPENING, 0.0584 + 0.045])
    in_hand[l == LINK_ID["panda_rightfingertip"]] += np.array([0.0, -FINGER_OPENING, 0.0584 + 0.045])
    in_link8 = in_hand @ R_hand.T
    if frame == "panda_hand":
        out = in_hand
    elif frame == "panda_link8":
        out = in_link8
    else:
        # link8 -> right_gripper: xyz (0,0,0.1), Rz(3pi/4)
        R_g = np.array([[-s, -s, 0.0], [s, -s, 0.0], [0.0, 0.0, 1.0]])
        out = (in_link8 - np.array([0.0, 0.0, 0.1])) @ R_g
    n = len(out)
    idx = (np.arange(total_points) * n) // total_points
    return np.ascontiguousarray(out[idx], dtype=np.float32)
