"""Development check of the split-bf16 training GEMMs against float64 (dense and sparse operands)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "motion-policy-networks_amd")]
import torch
from mpinets_amd import _lib
from mpinets_amd.pointnet2 import split_pairs

dev = torch.device("cuda:0")
torch.manual_seed(0)
M, N, K = 5000, 128, 256
for sparse in (False, True):
    x = torch.randn(M, K, device=dev)
    if sparse:
        x = x * (torch.rand(M, K, device=dev) < 0.02)
    w = torch.randn(N, K, device=dev) / 16
    mask = torch.randn(M, N, device=dev)
    y = torch.empty(M, N, device=dev)
    wp = split_pairs(w)
    _lib.call("mpx_linear_bf16x3_dact", _lib.ptr(x), K, _lib.ptr(wp), M, N, K, _lib.ptr(mask), N, 1, _lib.ptr(y), N)
    ref = (x.double() @ w.double().t()) * (mask > 0)
    y0 = torch.empty(M, N, device=dev)
    _lib.call("mpx_linear_bf16x3", _lib.ptr(x), K, _lib.ptr(wp), None, M, N, K, 0, _lib.ptr(y0), N)
    ref0 = x.double() @ w.double().t()
    print("sparse" if sparse else "dense", "dact rel err %.2e, plain rel err %.2e" % (
        ((y.double() - ref).abs().max() / ref.abs().max()).item(), ((y0.double() - ref0).abs().max() / ref0.abs().max()).item()))
    # wgrad: dW [N, K] = dz^T x
    dz = torch.randn(M, N, device=dev)
    if sparse:
        dz = dz * (torch.rand(M, N, device=dev) < 0.02)
    both = torch.empty(N * K + N, device=dev)
    scratch = torch.empty(_lib.load().mpx_linear_wgrad_scratch(M, N, K), device=dev)
    for fn in ("mpx_linear_wgrad", "mpx_linear_wgrad_bf16x3"):
        _lib.call(fn, _lib.ptr(dz), N, _lib.ptr(x), K, M, N, K, _lib.ptr(both), _lib.ptr(both[N * K:]), _lib.ptr(scratch))
        dw, db = both[:N * K].view(N, K), both[N * K:]
        rw, rb = dz.double().t() @ x.double(), dz.double().sum(0)
        print("   ", fn, "dW rel err %.2e, db rel err %.2e" % (((dw.double() - rw).abs().max() / rw.abs().max()).item(),
                                                              ((db.double() - rb).abs().max() / rb.abs().max()).item()))
