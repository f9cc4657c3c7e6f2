"""Policy forward: Python orchestration (mpinets_amd.model) vs the single C call (mpx_policy_forward).
(development aid)  usage: native_forward_timing.py B [B ...]"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "motion-policy-networks_amd")]
import torch

from mpinets_amd import _lib
from mpinets_amd.model import MotionPolicyNetwork
from mpinets_amd.scenes import make_problem_batch

dev = torch.device("cuda:0")
torch.manual_seed(0)
mdl = MotionPolicyNetwork().to(dev).eval()
for B in [int(a) for a in sys.argv[1:]] or [1, 256]:
    prob = make_problem_batch(B, seed=0, device=dev, scene_pool=64, device_clouds=True)
    xyz, q = prob["xyz"], prob["q_norm"]
    w, keep = mdl.native_weights()
    need = _lib.load().mpx_policy_workspace(B, xyz.size(1))
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    dq = torch.empty((B, 7), device=dev)

    def native():
        _lib.call("mpx_policy_forward", ctypes.addressof(w), _lib.ptr(xyz), xyz.size(1), _lib.ptr(q), B, _lib.ptr(dq), _lib.ptr(ws), need)

    def python():
        with torch.no_grad():
            mdl(xyz, q)

    res = {}
    for name, fn in (("python", python), ("native", native)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            fn()
        t_host = (time.perf_counter() - t0) / 50 * 1e3
        torch.cuda.synchronize()
        res[name] = ((time.perf_counter() - t0) / 50 * 1e3, t_host)
    print(f"B={B}: forward python {res['python'][0]:.3f} ms (host enqueue {res['python'][1]:.3f}), "
          f"one C call {res['native'][0]:.3f} ms (host {res['native'][1]:.3f}); workspace {need / 2**20:.1f} MiB")
