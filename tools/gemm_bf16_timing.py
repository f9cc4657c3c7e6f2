"""bf16x3 dense layers at the model's shapes: fp32-row input (split while staged) vs plane input (development aid).
usage: gemm_bf16_timing.py [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "motion-policy-networks_amd")]
import torch
from mpinets_amd import _lib
from mpinets_amd.pointnet2 import SplitWeights, linear_x3

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device("cuda:0")
split = SplitWeights()


def t(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def planes_of(x):
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    return hi.contiguous(), lo.contiguous()


torch.manual_seed(0)
for name, M, N, K, pool in [("sa3_l1", B * 128, 512, 272, 0), ("sa3_l2", B * 128, 512, 512, 0), ("sa3_l3", B * 128, 1024, 512, 1),
                            ("fc1", B, 4096, 1024, 0), ("fc2", B, 2048, 4096, 0), ("fc3", B, 2048, 2048, 0)]:
    x = torch.randn(M, K, device=dev).relu_()
    w = torch.randn(N, K, device=dev) * 0.05
    b = torch.randn(N, device=dev)
    wh, wl = split.get(w)
    flop3 = 3 * 2 * M * N * K
    if pool:
        y = torch.empty(M // 128, N, device=dev)
        ms = t(lambda: _lib.call("mpx_linear_rowmax_bf16x3", _lib.ptr(x), K, _lib.ptr(wh), _lib.ptr(wl), _lib.ptr(b), M, N, K,
                                 128, _lib.ptr(y), N))
        ref = y.clone()
    else:
        y = torch.empty(M, N, device=dev)
        ms = t(lambda: linear_x3(x, w, b, 1, split, out=y))
        ref = y.clone()
    line = f"{name:7s} M={M:8d} N={N:5d} K={K:5d}: rows-in {ms:7.3f} ms {flop3 / ms / 1e9:7.1f} TF(bf16)"
    if K % 32 == 0:
        xh, xl = planes_of(x)
        if pool:
            y2 = torch.empty_like(y)
            ms2 = t(lambda: _lib.call("mpx_linear_rowmax_bf16x3_planes", _lib.ptr(xh), _lib.ptr(xl), K, _lib.ptr(wh), _lib.ptr(wl),
                                      _lib.ptr(b), M, N, K, 128, _lib.ptr(y2), N))
            same = torch.equal(y2, ref)
            line += f" | planes-in {ms2:7.3f} ms {flop3 / ms2 / 1e9:7.1f} TF same={same}"
        else:
            y2 = torch.empty_like(y)
            ms2 = t(lambda: _lib.call("mpx_linear_bf16x3_planes", _lib.ptr(xh), _lib.ptr(xl), K, _lib.ptr(wh), _lib.ptr(wl),
                                      _lib.ptr(b), M, N, K, 1, _lib.ptr(y2), N, None, None, 0))
            same = torch.equal(y2, ref)
            ph = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            pl = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            ms3 = t(lambda: _lib.call("mpx_linear_bf16x3_planes", _lib.ptr(xh), _lib.ptr(xl), K, _lib.ptr(wh), _lib.ptr(wl),
                                      _lib.ptr(b), M, N, K, 1, None, 0, _lib.ptr(ph), _lib.ptr(pl), N))
            rh, rl = planes_of(ref)
            same3 = torch.equal(ph, rh) and torch.equal(pl, rl)
            line += f" | planes-in {ms2:7.3f} ms {flop3 / ms2 / 1e9:7.1f} TF same={same} | planes-in/out {ms3:7.3f} ms same={same3}"
    else:
        ph = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        pl = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        ms3 = t(lambda: _lib.call("mpx_linear_bf16x3_to_planes", _lib.ptr(x), K, _lib.ptr(wh), _lib.ptr(wl), _lib.ptr(b), M, N, K,
                                  1, _lib.ptr(ph), _lib.ptr(pl), N))
        rh, rl = planes_of(ref)
        line += f" | rows-in/planes-out {ms3:7.3f} ms same={torch.equal(ph, rh) and torch.equal(pl, rl)}"
    print(line, flush=True)
