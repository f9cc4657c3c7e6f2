"""Times the group-all module in bf16x3 at B environments: the layer-by-layer kernels against the fused front kernel + last
layer (tests/test_gpu_sa3_front.py holds the numerics).  usage: python tools/sa3_front_timing.py [B] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "motion-policy-networks_amd"), os.path.join(ROOT, "tests")]
import torch
from mpinets_amd import _lib
from mpinets_amd.pointnet2 import split_pairs
import test_gpu_sa3_front as t

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
K1, KR, C1, C2, C3 = t.K1, t.KR, t.C1, t.C2, t.C3
x, w, b = t._operands(8, 0)
x = x.to(dev).repeat(B // 8, 1).contiguous()
w = [v.to(dev).contiguous() for v in w]
b = [v.to(dev) for v in b]
M = B * 128
lib = _lib.load()
pack = torch.empty(lib.mpx_sa3_front_bf16x3_pack_size(K1, C1, C2), dtype=torch.uint8, device=dev)
_lib.call("mpx_sa3_front_bf16x3_pack", _lib.ptr(w[0]), KR, _lib.ptr(b[0]), _lib.ptr(w[1]), _lib.ptr(b[1]), K1, C1, C2, _lib.ptr(pack))
w3p = torch.empty((C3, 2 * C2), dtype=torch.bfloat16, device=dev)
_lib.call("mpx_sa3_front_bf16x3_w3_pairs", _lib.ptr(w[2]), C3, C2, _lib.ptr(w3p))
w0 = torch.nn.functional.pad(w[0], (0, K1 - KR)).contiguous()
wp = [split_pairs(w0), split_pairs(w[1]), split_pairs(w[2])]
p1 = torch.empty((M, 2 * C1), dtype=torch.bfloat16, device=dev)
p2 = torch.empty((M, 2 * C2), dtype=torch.bfloat16, device=dev)
pooled = torch.empty((B, C3), dtype=torch.float32, device=dev)

def l1(): _lib.call("mpx_linear_bf16x3_to_pairs", _lib.ptr(x), K1, _lib.ptr(wp[0]), _lib.ptr(b[0]), M, C1, K1, 1, _lib.ptr(p1), 2 * C1)
def l2(): _lib.call("mpx_linear_bf16x3_pairs", _lib.ptr(p1), 2 * C1, _lib.ptr(wp[1]), _lib.ptr(b[1]), M, C2, C1, 1, None, 0, _lib.ptr(p2), 2 * C2)
def l3(wpairs): _lib.call("mpx_linear_rowmax_bf16x3_pairs", _lib.ptr(p2), 2 * C2, _lib.ptr(wpairs), _lib.ptr(b[2]), M, C3, C2, 128, _lib.ptr(pooled), C3, None, 0)
def front(): _lib.call("mpx_sa3_front_bf16x3", _lib.ptr(x), K1, B, 128, _lib.ptr(pack), _lib.ptr(p2), 2 * C2)

def timed(f):
    f(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, c in ev:
        a.record(); f(); c.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(c) for a, c in ev)
    return ts[len(ts) // 2]

res = {"B": B, "layer1_ms": timed(l1), "layer2_ms": timed(l2), "layer3_ms": timed(lambda: l3(wp[2])), "front_ms": timed(front),
       "layer3_after_front_ms": timed(lambda: l3(w3p))}
macs12 = M * (K1 * C1 + C1 * C2)
res["front_pflops_bf16"] = 6 * macs12 / (res["front_ms"] * 1e-3) / 1e15
res["front_frac_of_2.5PF"] = res["front_pflops_bf16"] / 2.5
print(res)

if B > 300:
    import numpy as np
    probe = torch.zeros(64 + 4 * B, dtype=torch.int64, device=dev)
    _lib.call("mpx_sa3_front_bf16x3_probe", _lib.ptr(x), K1, B, _lib.ptr(pack), _lib.ptr(p2), 2 * C2, _lib.ptr(probe))
    torch.cuda.synchronize()
    tt = probe.cpu().numpy()
    n = int((tt[:8] != 0).sum())
    print("deltas of workgroup 300:", [int(tt[i + 1] - tt[i]) for i in range(n - 1)])
    w = tt[64:].reshape(B, 4)
    w = w[w[:, 0] > 0]  # (rows of workgroups that did not report are zero)
    t0 = w[:, 0].min()
    start, end = w[:, 0] - t0, w[:, 1] - t0
    dur = end - start
    cu = (w[:, 3] & 0xf) * 10000 + ((w[:, 2] >> 13) & 7) * 1000 + ((w[:, 2] >> 12) & 1) * 100 + ((w[:, 2] >> 8) & 15)
    print("workgroups:", len(w), "distinct (xcc, se, sh, cu):", len(set(cu.tolist())),
          "ticks per workgroup min / median / max:", int(dur.min()), int(np.median(dur)), int(dur.max()))
    gaps = []  # (s_memtime bases differ between XCDs: only differences inside one CU mean anything)
    for c in sorted(set(cu.tolist())):
        m = cu == c
        o = np.argsort(start[m])
        gaps += list(start[m][o][1:] - end[m][o][:-1])
    gaps = np.array(gaps)
    print("workgroups per CU", len(w) / len(set(cu.tolist())), "gap between consecutive workgroups of a CU min / median / max:",
          int(gaps.min()), int(np.median(gaps)), int(gaps.max()))
