"""GPU: the first set-abstraction module's two pooling forms agree bit for bit.

``sa_mlp_packed_kernel<1,64,64,64,...>`` pools through LDS float-max atomics when the output rows are 16-byte aligned and
a neighbourhood has <= 128 slots (round 6), else by the register merge with a flush at every query boundary; both evaluate
the same distinct neighbours with the same arithmetic, and max is exact -- so the pooled rows must be identical, also to the
kernel that walks all 128 slots (padding = copies of the first hit).  Reference: PointnetSAModule(npoint=512, radius=0.05,
nsample=128, mlp=[1, 64, 64, 64]) (/root/reference/mpinets/model.py:366-373)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,kinds", [(3, ("tabletop", "cubby", "dresser")), (16, ("tabletop",)), (5, ("dresser",)),
                                     (64, ("tabletop", "cubby")), (1040, ("tabletop", "cubby", "dresser"))])  # (64: 32 queries per unit; 1040: the persistent launch)
def test_lds_pooling_equals_register_merge_and_all_slots(B, kinds):
    from mpinets_amd import _lib
    from mpinets_amd.pointnet2 import PointnetSAModule, ball_query, furthest_point_sample, launch_sa
    from mpinets_amd.scenes import make_problem_batch

    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    sa = PointnetSAModule(npoint=512, radius=0.05, nsample=128, mlp=[1, 64, 64, 64], bn=False).to(dev)
    prob = make_problem_batch(B, seed=17 + B, device=dev, kinds=kinds, device_clouds=True, scene_pool=64)
    pc = prob["xyz"]
    # one dense cluster so that some neighbourhoods are FULL (128 of 128 slots, several tiles of one query)
    pc[0, 2048:2048 + 600, :3] = pc[0, 2048, :3] + 0.01 * torch.rand(600, 3, device=dev)
    N = pc.size(1)
    idx, new_xyz = furthest_point_sample(pc[:, :, :3].contiguous(), 512, return_xyz=True)
    nbr, cnt = ball_query(0.05, 128, pc[:, :, :3].contiguous(), new_xyz, return_counts=True)
    assert int(cnt.max()) == 128 and int(cnt.min()) >= 1
    convs = sa.convs()
    w = sa._packed.get(convs, 1, "fp32")
    widths = (64, 64, 64)

    def run(stride, counts):
        buf = torch.full((B, 512, stride), float("nan"), dtype=torch.float32, device=dev)
        launch_sa("fp32", _lib.ptr(pc), 4, _lib.ptr(new_xyz), 3, _lib.ptr(pc) + 12, 4, 1, nbr, counts, B, N, 512, 128, w, widths,
                  _lib.ptr(buf), stride)
        return buf[:, :, :64].clone()

    lds = run(68, cnt)        # aligned rows (the engine's [f1 | xyz | 0] layout): LDS pooling
    merge = run(65, cnt)      # rows that are not 16-byte aligned: the register merge
    slots = run(68, None)     # no counts: every slot of every neighbourhood
    assert torch.isfinite(lds).all()
    assert torch.equal(lds, merge)
    assert torch.equal(lds, slots)
