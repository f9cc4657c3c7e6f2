import os, sys
sys.path[:0] = ["/root/repo", "/root/repo/motion-policy-networks_amd"]
import numpy as np, torch
from mpinets_amd.pointnet2 import PointnetSAModule
dev = torch.device("cuda:0")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for (npoint, nsample, radius) in [(37, 64, 0.1)]:
    torch.manual_seed(3)
    mod = PointnetSAModule(npoint=npoint, radius=radius, nsample=nsample, mlp=[1, 64, 64, 64], bn=False).to(dev)
    rng = np.random.default_rng(npoint)
    xyz = (rng.uniform(-1, 1, (3, 700, 3)) * 0.5).astype(np.float32)
    feat = rng.normal(size=(3, 1, 700)).astype(np.float32)
    with torch.no_grad():
        nx, nf = mod(T(xyz), T(feat))
        mod.elide_padding = False
        _, nf_all = mod(T(xyz), T(feat))
    bad = (nf != nf_all)  # [B, C, npoint]
    print(npoint, nsample, radius, "mismatching (env, query):", sorted(set((int(b), int(q)) for b, c, q in bad.nonzero().tolist()))[:40],
          "channels bad per query:", bad.sum(dim=1).flatten().tolist()[:120])
    from mpinets_amd.pointnet2 import ball_query
