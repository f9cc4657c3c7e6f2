"""GPU: the fused first two layers of the group-all module in ``bf16x3`` (csrc/sa3_front_bf16.hip, mpx_sa3_front_bf16x3) against
a float64 evaluation of the same layers and against the layer-by-layer ``bf16x3`` kernels it replaces.

Reference semantics: PointnetSAModule(mlp=[256(+3), 512, 512, 1024]), npoint = None (model.py:377-383): three 1x1 convolutions
with ReLU over the 128 rows [xyz2 | f2] of an environment, max over the rows.  Tolerances: the split-bf16 arithmetic carries
~2^-17 relative error per product; the fused kernel must sit at the same distance from float64 as the kernels it replaces."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
K1, KR, C1, C2, C3 = 272, 259, 512, 512, 1024


def _operands(B, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.zeros((B * 128, K1), dtype=torch.float32)
    x[:, :3] = torch.rand((B * 128, 3), generator=g) * 2 - 1          # xyz2
    x[:, 3:KR] = torch.relu(torch.randn((B * 128, KR - 3), generator=g))  # pooled features are >= 0
    w = [torch.randn((C1, KR), generator=g) / KR ** 0.5, torch.randn((C2, C1), generator=g) / C1 ** 0.5,
         torch.randn((C3, C2), generator=g) / C2 ** 0.5]
    b = [torch.randn(n, generator=g) * 0.1 for n in (C1, C2, C3)]
    return x, w, b


def _reference(x, w, b, B):
    h = x[:, :KR].double()
    hs = []
    for wi, bi in zip(w, b):
        h = torch.relu(h @ wi.double().T + bi.double())
        hs.append(h)
    return hs[1], hs[2].view(B, 128, -1).max(dim=1).values


def _front(x, w, b, B):
    """mpx_sa3_front_bf16x3 -> (layer-2 rows in the kernel's channel order as fp32, pooled rows through the last layer)."""
    from mpinets_amd import _lib

    dev = x.device
    lib = _lib.load()
    n = lib.mpx_sa3_front_bf16x3_pack_size(K1, C1, C2)
    assert n > 0
    pack = torch.empty(n, dtype=torch.uint8, device=dev)
    _lib.call("mpx_sa3_front_bf16x3_pack", _lib.ptr(w[0]), KR, _lib.ptr(b[0]), _lib.ptr(w[1]), _lib.ptr(b[1]), K1, C1, C2, _lib.ptr(pack))
    w3p = torch.empty((C3, 2 * C2), dtype=torch.bfloat16, device=dev)
    _lib.call("mpx_sa3_front_bf16x3_w3_pairs", _lib.ptr(w[2]), C3, C2, _lib.ptr(w3p))
    p2 = torch.empty((B * 128, 2 * C2), dtype=torch.bfloat16, device=dev)
    _lib.call("mpx_sa3_front_bf16x3", _lib.ptr(x), K1, B, 128, _lib.ptr(pack), _lib.ptr(p2), 2 * C2)
    pooled = torch.empty((B, C3), dtype=torch.float32, device=dev)
    _lib.call("mpx_linear_rowmax_bf16x3_pairs", _lib.ptr(p2), 2 * C2, _lib.ptr(w3p), _lib.ptr(b[2]), B * 128, C3, C2, 128,
              _lib.ptr(pooled), C3, None, 0)
    # pairs -> fp32 rows in natural channel order: group s = [hi x 16 | lo x 16], position p <- channel kperm(s, p)
    q = p2.view(B * 128, C2 // 16, 2, 16).float()
    val = (q[:, :, 0] + q[:, :, 1]).reshape(B * 128, C2)  # position order
    s, p = torch.arange(C2) // 16, torch.arange(C2) % 16
    perm = 32 * (s // 2) + 16 * (s % 2) + (p % 4) + 8 * ((p // 4) % 2) + 4 * (p // 8)
    assert sorted(perm.tolist()) == list(range(C2))
    nat = torch.empty_like(val)
    nat[:, perm.to(dev)] = val
    return nat, pooled


def _layerwise(x, w, b, B):
    from mpinets_amd import _lib
    from mpinets_amd.pointnet2 import split_pairs

    dev = x.device
    w0 = torch.nn.functional.pad(w[0], (0, K1 - KR)).contiguous()
    wp = [split_pairs(w0), split_pairs(w[1]), split_pairs(w[2])]
    M = B * 128
    p1 = torch.empty((M, 2 * C1), dtype=torch.bfloat16, device=dev)
    p2 = torch.empty((M, 2 * C2), dtype=torch.bfloat16, device=dev)
    pooled = torch.empty((B, C3), dtype=torch.float32, device=dev)
    _lib.call("mpx_linear_bf16x3_to_pairs", _lib.ptr(x), K1, _lib.ptr(wp[0]), _lib.ptr(b[0]), M, C1, K1, 1, _lib.ptr(p1), 2 * C1)
    _lib.call("mpx_linear_bf16x3_pairs", _lib.ptr(p1), 2 * C1, _lib.ptr(wp[1]), _lib.ptr(b[1]), M, C2, C1, 1, None, 0, _lib.ptr(p2), 2 * C2)
    _lib.call("mpx_linear_rowmax_bf16x3_pairs", _lib.ptr(p2), 2 * C2, _lib.ptr(wp[2]), _lib.ptr(b[2]), M, C3, C2, 128,
              _lib.ptr(pooled), C3, None, 0)
    q = p2.view(M, C2 // 16, 2, 16).float()
    return (q[:, :, 0] + q[:, :, 1]).reshape(M, C2), pooled


@pytest.mark.parametrize("B,seed", [(1, 0), (3, 1), (40, 2)])
def test_fused_front_against_float64_and_the_layerwise_kernels(B, seed):
    dev = torch.device("cuda:0")
    x, w, b = _operands(B, seed)
    ref2, ref_pool = _reference(x, w, b, B)
    xd, wd, bd = x.to(dev), [t.to(dev).contiguous() for t in w], [t.to(dev) for t in b]
    h2_f, pool_f = _front(xd, wd, bd, B)
    h2_l, pool_l = _layerwise(xd, wd, bd, B)
    torch.cuda.synchronize()
    scale2, scale3 = ref2.abs().max().item(), ref_pool.abs().max().item()
    e = {"front h2": (h2_f.cpu().double() - ref2).abs().max().item() / scale2,
         "layerwise h2": (h2_l.cpu().double() - ref2).abs().max().item() / scale2,
         "front pooled": (pool_f.cpu().double() - ref_pool).abs().max().item() / scale3,
         "layerwise pooled": (pool_l.cpu().double() - ref_pool).abs().max().item() / scale3,
         "front vs layerwise pooled": (pool_f - pool_l).abs().max().item() / scale3}
    print("relative to the largest entry:", {k: "%.2e" % v for k, v in e.items()})
    assert e["front h2"] <= 1.5e-5 and e["front pooled"] <= 1.5e-5, e   # (rows read back from hi + lo pairs: 2^-17 relative at best)
    assert e["front h2"] <= 1.5 * e["layerwise h2"] + 1e-7 and e["front pooled"] <= 1.5 * e["layerwise pooled"] + 1e-7, e


def test_rejects_other_shapes():
    from mpinets_amd import _lib

    assert _lib.load().mpx_sa3_front_bf16x3_pack_size(256, 512, 512) == -1
    x = torch.zeros((128, K1), device="cuda:0")
    with pytest.raises(_lib.MpxError):
        _lib.call("mpx_sa3_front_bf16x3", _lib.ptr(x), K1, 1, 64, _lib.ptr(x), _lib.ptr(x), 1024)
