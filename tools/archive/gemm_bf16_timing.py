"""bf16x3 dense layers at the model's shapes: fp32-row input (split while staged) vs pairs input (development aid).
usage: gemm_bf16_timing.py [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "motion-policy-networks_amd")]
import torch
from mpinets_amd import _lib
from mpinets_amd.pointnet2 import SplitWeights, linear_x3, split_pairs

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


torch.manual_seed(0)
for name, M, N, K, pool in [("sa3_l1", B * 128, 512, 272, 0), ("sa3_l2", B * 128, 512, 512, 0), ("sa3_l3", B * 128, 1024, 512, 1),
                            ("fc1", B, 4096, 1024, 0), ("fc2", B, 2048, 4096, 0), ("fc3", B, 2048, 2048, 0)]:
    x = torch.randn(M, K, device=dev).relu_()
    w = torch.randn(N, K, device=dev) * 0.05
    b = torch.randn(N, device=dev)
    wp = split.get(w)
    xp = split_pairs(x)
    flop3 = 3 * 2 * M * N * K
    if pool:
        y = torch.empty(M // 128, N, device=dev)
        ms = t(lambda: _lib.call("mpx_linear_rowmax_bf16x3", _lib.ptr(x), K, _lib.ptr(wp), _lib.ptr(b), M, N, K, 128, _lib.ptr(y), N))
        ref = y.clone()
        y2 = torch.empty_like(y)
        ms2 = t(lambda: _lib.call("mpx_linear_rowmax_bf16x3_pairs", _lib.ptr(xp), 2 * K, _lib.ptr(wp), _lib.ptr(b), M, N, K, 128,
                                  _lib.ptr(y2), N, None, 0))
        print(f"{name:7s} M={M:8d} N={N:5d} K={K:5d}: rows-in {ms:7.3f} ms {flop3 / ms / 1e9:7.1f} TF(bf16) | pairs-in {ms2:7.3f} ms "
              f"{flop3 / ms2 / 1e9:7.1f} TF same={torch.equal(y2, ref)}", flush=True)
        continue
    y = torch.empty(M, N, device=dev)
    ms = t(lambda: linear_x3(x, w, b, 1, split, out=y))
    ref = y.clone()
    yp = torch.empty(M, 2 * N, dtype=torch.bfloat16, device=dev)
    ms1 = t(lambda: _lib.call("mpx_linear_bf16x3_to_pairs", _lib.ptr(x), K, _lib.ptr(wp), _lib.ptr(b), M, N, K, 1, _lib.ptr(yp), 2 * N))
    rp = split_pairs(ref)
    same1 = torch.equal(yp, rp)
    y2 = torch.empty_like(y)
    ms2 = t(lambda: _lib.call("mpx_linear_bf16x3_pairs", _lib.ptr(xp), 2 * K, _lib.ptr(wp), _lib.ptr(b), M, N, K, 1, _lib.ptr(y2), N, None, 0))
    ms3 = t(lambda: _lib.call("mpx_linear_bf16x3_pairs", _lib.ptr(xp), 2 * K, _lib.ptr(wp), _lib.ptr(b), M, N, K, 1, None, 0, _lib.ptr(yp), 2 * N))
    print(f"{name:7s} M={M:8d} N={N:5d} K={K:5d}: rows-in {ms:7.3f} ms {flop3 / ms / 1e9:7.1f} TF(bf16) | rows->pairs {ms1:7.3f} same={same1}"
          f" | pairs-in {ms2:7.3f} ms {flop3 / ms2 / 1e9:7.1f} TF same={torch.equal(y2, ref)} | pairs->pairs {ms3:7.3f} ms "
          f"same={torch.equal(yp, rp)}", flush=True)
