import os, sys
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "motion-policy-networks_amd")]
import torch
from mpinets_amd.pointnet2 import groupnorm_leaky_train
dev = torch.device("cuda:0")
torch.manual_seed(3)
for (M, C) in [(7, 4096), (256, 2048), (64, 2048), (65, 2048), (256, 64)]:
    gn = torch.nn.GroupNorm(16, C).to(dev)
    with torch.no_grad():
        gn.weight.normal_(1.0, 0.3), gn.bias.normal_(0.0, 0.3)
    x = (torch.randn(M, C, device=dev) * 2 + 0.5).requires_grad_(True)
    g = torch.randn(M, C, device=dev)
    (groupnorm_leaky_train(x, gn) * g).sum().backward()
    got = (x.grad.clone(), gn.weight.grad.clone(), gn.bias.grad.clone())
    for where in ("cuda:0", "cpu"):
        xd = x.detach().double().to(where).requires_grad_(True)
        gd = torch.nn.GroupNorm(16, C).double().to(where)
        gd.load_state_dict({k: v.double().to(where) for k, v in gn.state_dict().items()})
        (torch.nn.functional.leaky_relu(gd(xd), 0.01) * g.double().to(where)).sum().backward()
        errs = [(a.double().cpu() - b.cpu()).abs().max().item() for a, b in zip(got, (xd.grad, gd.weight.grad, gd.bias.grad))]
        print(M, C, where, "dx %.2e dgamma %.2e dbeta %.2e" % tuple(errs))
