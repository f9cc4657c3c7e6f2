#!/usr/bin/env python
"""Dump robofin's robot tables into the .npz the engine's samplers accept.  Run it WHERE ROBOFIN IS INSTALLED
(the reference's docker image, docker/Dockerfile:153 pins robofin v0.0.1); it is not needed -- and cannot run --
in the build image, which has no robofin and no meshes.

    python tools/dump_robofin_tables.py robofin_tables.npz

writes  points [P,3] float32 (link-local), point_link_name [P] (URDF link names), joint_limits_real [7,2],
joint_limits_published [7,2];  use it with

    from mpinets_amd import franka_tables as ft
    tabs = ft.load_point_tables("robofin_tables.npz")
    sampler = FrankaSampler(device, point_table=tabs["point_table"])

What it reads (robofin v0.0.1, [EXT-RECALL] -- the attribute names are probed, not assumed):
  robofin.robots.FrankaRealRobot.JOINT_LIMITS / FrankaRobot.JOINT_LIMITS      -> the two limit tables
  robofin.pointcloud.torch.FrankaSampler(device, use_cache=True).points        -> {link name: [1,n,3|4] mesh samples
                                                                                   in the link's VISUAL frame}
robofin transforms those samples with the visual-geometry FK (link frame x visual origin).  The Panda's visual
origins are identity in the public URDF, so the samples are taken as link-local; the self-check at the end compares
robofin's own sample(q) with this engine's FK applied to the dumped table and prints the worst deviation -- if it is
not ~1e-6 the visual origins are not identity in your URDF and the table must not be used as is.
"""
import sys

import numpy as np


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "robofin_tables.npz"
    try:
        import torch
        from robofin.pointcloud.torch import FrankaSampler as RobofinSampler
        from robofin.robots import FrankaRealRobot, FrankaRobot
    except ImportError as e:
        sys.exit(f"robofin is not importable here ({e}); run this where the reference's environment is installed")
    sampler = RobofinSampler("cpu", use_cache=True)
    tables = getattr(sampler, "points", None)
    if not isinstance(tables, dict) or not tables:
        sys.exit("this robofin version keeps its link point tables somewhere else than FrankaSampler.points: "
                 f"attributes are {sorted(vars(sampler))}")
    pts, names = [], []
    for link, t in tables.items():
        a = np.asarray(torch.as_tensor(t).detach().cpu().reshape(-1, torch.as_tensor(t).shape[-1])[:, :3], np.float32)
        if link.startswith("eef_"):  # (end-effector tables of sample_end_effector, if this version caches them here)
            continue
        pts.append(a)
        names += [link] * len(a)
    payload = dict(points=np.concatenate(pts), point_link_name=np.asarray(names, dtype="U32"),
                   joint_limits_real=np.asarray(FrankaRealRobot.JOINT_LIMITS, np.float64),
                   joint_limits_published=np.asarray(FrankaRobot.JOINT_LIMITS, np.float64))
    np.savez(out, **payload)
    print(f"{out}: {len(names)} points on {len(set(names))} links")
    # ---- self-check against robofin's own sampler (needs this engine's library and a GPU) ------------------
    try:
        import os

        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                        "motion-policy-networks_amd"))
        from mpinets_amd import franka_tables as ft
        from mpinets_amd.robot import FrankaSampler

        if not torch.cuda.is_available():
            raise RuntimeError("no GPU")
        tabs = ft.load_point_tables(out)
        mine = FrankaSampler("cuda:0", point_table=tabs["point_table"])
        q = torch.as_tensor(ft.DEFAULT_Q, dtype=torch.float32)[None]
        theirs = sampler.sample(q)[0, :, :3].numpy()  # robofin returns a random subset: compare as point sets
        ours = mine.sample(q.cuda())[0].cpu().numpy()
        from scipy.spatial import cKDTree

        d = cKDTree(ours).query(theirs)[0].max()
        print(f"self-check: robofin sample(default_q) vs engine FK x dumped table: worst nearest-point distance {d:.2e} m"
              + ("  (OK)" if d < 1e-5 else "  (MISMATCH: visual origins / FK constants differ -- do not use this table)"))
        lim = np.abs(payload["joint_limits_real"] - ft.JOINT_LIMITS_REAL).max()
        print(f"joint limits: |robofin FrankaRealRobot - franka_tables.JOINT_LIMITS_REAL| max = {lim:.2e}")
    except Exception as e:  # the dump itself is complete without it
        print(f"self-check skipped: {e}")


if __name__ == "__main__":
    main()
