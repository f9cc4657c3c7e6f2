"""Generates tests/golden/franka_golden.npz from the ONLY Franka data the reference tree holds:

* /root/reference/config/franka_robot_description.yaml:44-46 (default_q), :51-53 (finger joint values),
  :57-182 (collision spheres: 57 spheres on 11 links, 10 distinct radii) -- the sphere model
  FrankaCollisionSampler / model.py:293-314 works with;
* /root/reference/config/franka_fabric_config.yaml:117-140 (body cylinder + self-collision spheres used by
  mpx_trajectory_metrics' self-collision flag).

Run in the build container (the reference does not travel):  python tests/golden/gen_franka_golden.py
The fixture is DATA (numbers and link names parsed from the two YAML files), not reference source text.
"""
import os

import numpy as np
import yaml

REF = "/root/reference/config"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "franka_golden.npz")


def main():
    desc = yaml.safe_load(open(os.path.join(REF, "franka_robot_description.yaml")))
    links, centers, radii = [], [], []
    for entry in desc["collision_spheres"]:
        (link, spheres), = entry.items()
        for s in spheres:
            links.append(link)
            centers.append(s["center"])
            radii.append(s["radius"])
    fingers = {r["name"]: r["value"] for r in desc["cspace_to_urdf_rules"]}
    fab = yaml.safe_load(open(os.path.join(REF, "franka_fabric_config.yaml")))
    cyl = fab["body_cylinders"][0]
    np.savez(
        OUT,
        sphere_link=np.asarray(links, dtype="U32"), sphere_center=np.asarray(centers, np.float64),
        sphere_radius=np.asarray(radii, np.float64), default_q=np.asarray(desc["default_q"], np.float64),
        cspace=np.asarray(desc["cspace"], dtype="U32"),
        finger_joint=np.asarray(sorted(fingers), dtype="U32"),
        finger_value=np.asarray([fingers[k] for k in sorted(fingers)], np.float64),
        body_cylinder_pt1=np.asarray(cyl["pt1"], np.float64), body_cylinder_pt2=np.asarray(cyl["pt2"], np.float64),
        body_cylinder_radius=np.float64(cyl["radius"]),
        self_sphere_name=np.asarray([s["name"] for s in fab["self_collision_spheres"]], dtype="U32"),
        self_sphere_radius=np.asarray([s["radius"] for s in fab["self_collision_spheres"]], np.float64),
    )
    print(f"{OUT}: {len(radii)} spheres on {len(set(links))} links, {len(set(radii))} radii")


if __name__ == "__main__":
    main()
