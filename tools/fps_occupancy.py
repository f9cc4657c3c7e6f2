import os, sys
sys.path[:0] = ['.', 'motion-policy-networks_amd']
import torch
from mpinets_amd import _lib
from mpinets_amd.scenes import make_problem_batch
dev = torch.device("cuda:0")
N = int(sys.argv[1]); NP = 512
for B in (256, 512, 768, 1024, 1536, 2048):
    prob = make_problem_batch(B, seed=0, device=dev, scene_pool=64, device_clouds=True)
    perm = torch.randperm(6272, device=dev)[:N].sort().values
    xyz = prob["xyz"][:, perm].contiguous()
    idx = torch.empty((B, NP), dtype=torch.int32, device=dev)
    def run():
        _lib.call("mpx_fps", _lib.ptr(xyz), B, N, 4, NP, _lib.ptr(idx), None, 3)
    for _ in range(2): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    print(f"cull={os.environ.get('MPX_FPS_CULL','1')} N={N} B={B}: {e0.elapsed_time(e1)/5:.3f} ms", flush=True)
