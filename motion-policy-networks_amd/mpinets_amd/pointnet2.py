"""PointNet++ op seam: drop-in for ``pointnet2_ops`` as the reference uses it.

Reference call sites: ``from pointnet2_ops.pointnet2_modules import PointnetSAModule``
(mpinets/model.py:27), constructed at model.py:366-383, called at model.py:423-424 with
``xyz [B,N,3]`` contiguous and ``features [B,C,N]`` contiguous; returns
``(new_xyz [B,npoint,3] | None, new_features [B,mlp[-1],npoint])``.  xyz is concatenated before
the features (3 extra input channels), ``bn=False`` gives Conv2d(bias=True)+ReLU per layer, and the
state-dict keys are ``mlps.0.{0,2,4}.{weight,bias}`` ([EXT-RECALL]).

All compute runs in ``libmpinets_hip.so``; CPU tensors are rejected like in the reference
(model.py:417 "CPU tensors not supported").
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
from torch import nn

from . import _lib


# ---- functional ops (pointnet2_utils equivalents) --------------------------------------------------
def furthest_point_sample(xyz: torch.Tensor, npoint: int, return_xyz: bool = False):
    """xyz [B,N,3|4] float32 (rows may be slab rows: only the first three columns are read)
    -> idx int32 [B,npoint] (and new_xyz [B,npoint,3])."""
    _lib.require_cuda(xyz)
    assert xyz.ndim == 3 and xyz.size(2) >= 3 and xyz.dtype == torch.float32 and xyz.is_contiguous()
    B, N, S = xyz.shape
    idx = torch.empty((B, npoint), dtype=torch.int32, device=xyz.device)
    new_xyz = torch.empty((B, npoint, 3), dtype=torch.float32, device=xyz.device) if return_xyz else None
    _lib.call("mpx_fps", _lib.ptr(xyz), B, N, S, npoint, _lib.ptr(idx), _lib.ptr(new_xyz), 3)
    return (idx, new_xyz) if return_xyz else idx


def gather_operation(features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """features [B,C,N], idx [B,npoint] -> [B,C,npoint] (index plumbing, torch)."""
    return torch.gather(features, 2, idx.long().unsqueeze(1).expand(-1, features.size(1), -1))


def ball_query(radius: float, nsample: int, xyz: torch.Tensor, new_xyz: torch.Tensor, return_counts: bool = False):
    """-> idx int32 [B,npoint,nsample] (argument order of pointnet2_utils.ball_query); with
    ``return_counts`` also the number of real hits per query (slots beyond it are padding)."""
    _lib.require_cuda(xyz, new_xyz)
    assert xyz.is_contiguous() and new_xyz.is_contiguous()
    B, N, S = xyz.shape
    npoint = new_xyz.size(1)
    idx = torch.empty((B, npoint, nsample), dtype=torch.int32, device=xyz.device)
    cnt = torch.empty((B, npoint), dtype=torch.int32, device=xyz.device) if return_counts else None
    _lib.call("mpx_ball_query", _lib.ptr(new_xyz), new_xyz.size(2), _lib.ptr(xyz), S, B, N, npoint,
              float(radius), nsample, _lib.ptr(idx), _lib.ptr(cnt))
    return (idx, cnt) if return_counts else idx


def query_and_group(xyz: torch.Tensor, new_xyz: torch.Tensor, features_pm: Optional[torch.Tensor],
                    idx: torch.Tensor) -> torch.Tensor:
    """Materialised QueryAndGroup(use_xyz=True): -> [B,3+C,npoint,nsample].
    ``features_pm`` is point-major [B,N,C]."""
    B, N, S = xyz.shape
    npoint, nsample = idx.shape[1:]
    C = 0 if features_pm is None else features_pm.size(2)
    out = torch.empty((B, 3 + C, npoint, nsample), dtype=torch.float32, device=xyz.device)
    _lib.call("mpx_group_points", _lib.ptr(xyz), S, _lib.ptr(new_xyz), new_xyz.size(2), _lib.ptr(features_pm),
              C if features_pm is None else features_pm.stride(1), C, _lib.ptr(idx), B, N, npoint, nsample,
              _lib.ptr(out))
    return out


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], act: int = 0,
           out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = act(x @ weight.T + bias) on the fp32 matrix cores.  x [M,K] (row stride may exceed K),
    weight [N,K].  ``out`` may be a column slice of a wider row-major buffer."""
    assert x.ndim == 2 and weight.ndim == 2 and x.size(1) == weight.size(1)
    assert x.stride(1) == 1 and weight.is_contiguous()
    M, K = x.shape
    N = weight.size(0)
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    assert out.shape == (M, N) and out.stride(1) == 1
    need = _lib.load().mpx_linear_workspace(M, N, K)
    if need == 0:
        _lib.call("mpx_linear", _lib.ptr(x), x.stride(0), _lib.ptr(weight), _lib.ptr(bias), M, N, K, act,
                  _lib.ptr(out), out.stride(0))
    else:  # skinny problem: split K over the CUs through a per-(device, stream) workspace (stream-ordered reuse)
        key = (x.device, _lib.stream_ptr())
        ws = _WORKSPACE.get(key)
        if ws is None or ws.numel() < need:
            ws = _WORKSPACE[key] = torch.empty(max(need, 16 << 20), dtype=torch.uint8, device=x.device)
        _lib.call("mpx_linear_ws", _lib.ptr(x), x.stride(0), _lib.ptr(weight), _lib.ptr(bias), M, N, K, act,
                  _lib.ptr(out), out.stride(0), _lib.ptr(ws), ws.numel())
    return out


_WORKSPACE = {}


def split_pairs(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp32 rows [R, K] -> the ``bf16x3`` kernels' PAIRS form [R, 2 * roundup(K, 16)] bf16: per group of 16 k-values
    [hi x 16 | lo x 16], hi = bf16(v), lo = bf16(v - hi), zero padded."""
    assert x.ndim == 2 and x.stride(1) == 1 and x.dtype == torch.float32
    R, K = x.shape
    Kp = (K + 15) // 16 * 16
    if out is None:
        out = torch.empty((R, 2 * Kp), dtype=torch.bfloat16, device=x.device)
    assert out.shape == (R, 2 * Kp) and out.stride(1) == 1
    _lib.call("mpx_split_bf16", _lib.ptr(x), x.stride(0), R, K, _lib.ptr(out), out.stride(0))
    return out


class SplitWeights:
    """Dense weight matrices in the pairs form for the ``bf16x3`` mode, refreshed when a parameter changes."""

    def __init__(self):
        self.cache = {}

    def get(self, weight: torch.Tensor, source: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``source``: the parameter a derived matrix (padded / transposed copy) was made from -- its version
        decides when the pairs are stale."""
        src = weight if source is None else source
        key = (src.data_ptr(), tuple(weight.shape))
        ver = (src._version, tuple(weight.shape))
        hit = self.cache.get(key)
        if hit is None or hit[0] != ver:
            w = _lib.f32c(weight.detach())
            hit = (ver, split_pairs(w), w)  # w kept alive: its data_ptr is the key
            self.cache[key] = hit
        return hit[1]


def linear_x3(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], act: int, split: SplitWeights,
              out: Optional[torch.Tensor] = None, source: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``linear`` on the bf16 matrix cores (three split products per fp32 product, fp32 accumulate)."""
    assert x.ndim == 2 and weight.ndim == 2 and x.size(1) == weight.size(1) and x.stride(1) == 1
    M, K = x.shape
    N = weight.size(0)
    wp = split.get(weight, source)
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    assert out.shape == (M, N) and out.stride(1) == 1
    _lib.call("mpx_linear_bf16x3", _lib.ptr(x), x.stride(0), _lib.ptr(wp), _lib.ptr(bias), M, N, K, act, _lib.ptr(out),
              out.stride(0))
    return out


def groupnorm_leaky(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, groups: int, eps: float = 1e-5,
                    out: Optional[torch.Tensor] = None) -> torch.Tensor:
    assert x.ndim == 2 and x.is_contiguous()
    if out is None:
        out = torch.empty_like(x)
    _lib.call("mpx_groupnorm_leaky", _lib.ptr(x), _lib.ptr(weight), _lib.ptr(bias), x.size(0), x.size(1), groups,
              float(eps), _lib.ptr(out))
    return out


PRECISIONS = ("fp32", "bf16x3")
# training: the backward of [last dense layer + ReLU + max-pool] walks the pool's Q * C non-zero gradients
# (mpx_pool_wgrad / mpx_pool_dgrad) instead of forming the [rows, C] gradient and running two dense GEMMs over its zeros;
# False = the dense route (mpx_segment_max_grad_act + mpx_linear_wgrad / mpx_linear_dact), kept for A/B tests
SPARSE_POOL_BACKWARD = True
# training: the last layer's GEMM of a pooled stack max-pools in its epilogue (mpx_linear_segmax*): bit-identical to
# mpx_linear + mpx_segment_max, without the [rows, C] matrix in memory; False = the two-step form (A/B tests)
FUSED_POOL_FORWARD = True


class SAWeights:
    """Packed (MFMA-stream order) weights of one shared MLP, refreshed when parameters change.
    One pack per precision mode: ``fp32`` (exact fp32 MFMA, the parity default) or ``bf16x3``
    (split-bf16: hi/lo bf16 operand blocks for the bf16 matrix cores)."""

    def __init__(self):
        self.packs = {}
        self._fact = None

    def factored(self, convs: List[nn.Conv2d], C: int):
        """First layer split for ``mpx_sa_mlp_factored``: (w_point [c1, Kp] over rows [feat | xyz | 0],
        w_centre [c1, 4] over rows [xyz | 0], -b1)."""
        c0 = convs[0]
        ver = (c0.weight._version, c0.bias._version, c0.weight.data_ptr())
        if self._fact is None or self._fact[0] != ver:
            w = c0.weight.detach().reshape(c0.out_channels, -1).float()
            assert w.size(1) == 3 + C
            z = lambda n: torch.zeros((w.size(0), n), dtype=torch.float32, device=w.device)
            wp = torch.cat((w[:, 3:], w[:, :3], z((-(3 + C)) % 4)), dim=1).contiguous()
            wc = torch.cat((w[:, :3], z(1)), dim=1).contiguous()
            self._fact = (ver, (wp, wc, (-c0.bias.detach().float()).contiguous()))
        return self._fact[1]

    def get(self, convs: List[nn.Conv2d], C: int, precision: str = "fp32") -> torch.Tensor:
        assert precision in PRECISIONS, precision
        ver = tuple((c.weight._version, c.bias._version, c.weight.data_ptr()) for c in convs)
        hit = self.packs.get(precision)
        if hit is None or hit[0] != ver:
            c1, c2, c3 = (c.out_channels for c in convs)
            lib = _lib.load()
            dev = convs[0].weight.device
            w = [_lib.f32c(c.weight.detach().reshape(c.out_channels, -1)) for c in convs]
            b = [_lib.f32c(c.bias.detach()) for c in convs]
            if precision == "fp32":
                n = lib.mpx_sa_pack_size(C, c1, c2, c3)
                pack = torch.empty(max(n, 0), dtype=torch.float32, device=dev)
                fn = "mpx_sa_pack_weights"
            else:
                n = lib.mpx_sa_pack_bf16x3_size(C, c1, c2, c3)
                pack = torch.empty(max(n, 0), dtype=torch.uint8, device=dev)
                fn = "mpx_sa_pack_bf16x3"
            if n < 0:
                raise _lib.MpxError(f"unsupported shared-MLP shape C={C} mlp=({c1},{c2},{c3})")
            _lib.call(fn, _lib.ptr(w[0]), _lib.ptr(b[0]), _lib.ptr(w[1]), _lib.ptr(b[1]), _lib.ptr(w[2]),
                      _lib.ptr(b[2]), C, c1, c2, c3, _lib.ptr(pack))
            self.packs[precision] = (ver, pack)
            hit = self.packs[precision]
        return hit[1]


def launch_sa(precision: str, xyz_ptr: int, stride: int, new_xyz_ptr: int, new_stride: int, feat_ptr: int,
              feat_stride: int, C: int, idx: torch.Tensor, cnt: Optional[torch.Tensor], B: int, N: int, npoint: int,
              nsample: int, wpack: torch.Tensor, widths: Tuple[int, int, int], out_ptr: int, out_stride: int,
              append_centre: bool = False) -> bool:
    """One fused group + MLP + max-pool launch in either precision (raw pointers: slab views welcome).
    ``cnt`` (from the ball query) lets the kernel skip neighbourhood tiles that hold only padding --
    bit-identical output; for the lockstep bf16x3 kernel the queries are first ordered by tile count.
    ``append_centre``: also write [query xyz | 0] into columns [c3, c3+4) of the output rows (the fp32 kernel with
    counts and the weight-resident bf16x3 kernel); returns whether that was done (False: the caller appends them with ``mpx_append_columns``)."""
    c1, c2, c3 = widths
    if precision == "fp32":
        fused = bool(append_centre and cnt is not None)
        _lib.call("mpx_sa_mlp", xyz_ptr, stride, new_xyz_ptr, new_stride, feat_ptr, feat_stride, C, _lib.ptr(idx),
                  _lib.ptr(cnt), B, N, npoint, nsample, _lib.ptr(wpack), c1, c2, c3, out_ptr, out_stride, int(fused))
        return fused
    order = None
    wants_order = bool(_lib.load().mpx_sa_mlp_bf16x3_wants_order(C, c1, c2, c3, nsample))
    if cnt is not None and wants_order:
        order = torch.empty(B * npoint, dtype=torch.int32, device=idx.device)
        scratch = torch.empty(128, dtype=torch.int32, device=idx.device)
        _lib.call("mpx_sort_queries", _lib.ptr(cnt), B * npoint, nsample, _lib.ptr(order), _lib.ptr(scratch))
    fused = bool(append_centre and not wants_order)  # (the weight-resident kernel completes the rows itself)
    _lib.call("mpx_sa_mlp_bf16x3", xyz_ptr, stride, new_xyz_ptr, new_stride, feat_ptr, feat_stride, C, _lib.ptr(idx),
              _lib.ptr(cnt), _lib.ptr(order), B, N, npoint, nsample, _lib.ptr(wpack), c1, c2, c3, out_ptr, out_stride,
              int(fused))
    return fused


def sa_mlp_fused(xyz: torch.Tensor, new_xyz: torch.Tensor, feat: torch.Tensor, feat_stride: int, C: int,
                 idx: torch.Tensor, wpack: torch.Tensor, widths: Tuple[int, int, int],
                 out: Optional[torch.Tensor] = None, precision: str = "fp32",
                 cnt: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Fused group + MLP + max-pool.  xyz [B,N,S]; new_xyz [B,npoint,S']; feat: tensor whose
    ``data_ptr`` is the first feature of point 0 with ``feat_stride`` floats between points;
    -> out [B,npoint,c3] point-major (``out`` may be a column slice of a wider buffer)."""
    B, N, S = xyz.shape
    npoint, nsample = idx.shape[1:]
    c1, c2, c3 = widths
    if out is None:
        out = torch.empty((B, npoint, c3), dtype=torch.float32, device=xyz.device)
    assert out.stride(2) == 1 and out.stride(0) == npoint * out.stride(1)
    launch_sa(precision, _lib.ptr(xyz), S, _lib.ptr(new_xyz), new_xyz.stride(1), _lib.ptr(feat), feat_stride, C, idx, cnt,
              B, N, npoint, nsample, wpack, widths, _lib.ptr(out), out.stride(1))
    return out


def sa_mlp_factored(point_rows: torch.Tensor, centre_rows: torch.Tensor, idx: torch.Tensor, cnt: torch.Tensor,
                    packed: "SAWeights", convs: List[nn.Conv2d], C: int, N: int, out_ptr: int, out_stride: int,
                    precision: str = "fp32", split: Optional["SplitWeights"] = None) -> None:
    """The (64+3,128,128,256) module with its first layer evaluated per point / per query instead of per
    (query, neighbour) row (``mpx_sa_mlp_factored`` / ``mpx_sa_mlp_bf16x3_factored``).  ``point_rows`` [B*N, Kp] =
    [feat | xyz | 0] rows, ``centre_rows`` [B*npoint, 4] = [xyz | anything finite] rows (row strides free)."""
    B, npoint, nsample = idx.shape
    wp, wc, nb1 = packed.factored(convs, C)
    c1, c2, c3 = (c.out_channels for c in convs)
    if precision == "bf16x3":
        # (K = 68: HBM-bound either way -- the fp32 row-per-lane kernel is the faster one, 0.71 vs 0.79 ms, and exact)
        pre = linear(point_rows, wp, None)
        ctr = linear(centre_rows, wc, nb1)  # K = 4: not worth the matrix cores
        # (the factored kernel is persistent and takes its units from a device-side queue: no sorting pass, order = NULL)
        _lib.call("mpx_sa_mlp_bf16x3_factored", _lib.ptr(pre), _lib.ptr(ctr), _lib.ptr(idx), _lib.ptr(cnt), None,
                  B, N, npoint, nsample, _lib.ptr(packed.get(convs, C, "bf16x3")), C, c1, c2, c3, out_ptr, out_stride)
        return
    pre = linear(point_rows, wp, None)
    ctr = linear(centre_rows, wc, nb1)
    _lib.call("mpx_sa_mlp_factored", _lib.ptr(pre), _lib.ptr(ctr), _lib.ptr(idx), _lib.ptr(cnt), B, N, npoint, nsample,
              _lib.ptr(packed.get(convs, C, "fp32")), C, c1, c2, c3, out_ptr, out_stride)


FACTORED_SHAPE = (64, 128, 128, 256)
FACTORED_BF16X3_MAX_NSAMPLE = 128  # (= csrc/sa_mlp_bf16.hip v2::MAX_NSAMPLE: the persistent kernel's row -> query table)


def use_factored(module: "PointnetSAModule", C: int, convs) -> bool:
    """Does this module run with its first layer factored per point / per query?  Only the (64+3, 128, 128, 256) shape has
    the kernels, and the ``bf16x3`` one is built for <= 128 slots per neighbourhood (larger ones take the unfactored
    kernel, which has no such limit)."""
    if not module.factored or (C,) + tuple(c.out_channels for c in convs) != FACTORED_SHAPE:
        return False
    return module.precision != "bf16x3" or module.nsample <= FACTORED_BF16X3_MAX_NSAMPLE


# ---- module -------------------------------------------------------------------------------------------
# ---- training path (row N1): dense layers with hand-written forward AND backward kernels ------------------------
class _LinearFn(torch.autograd.Function):
    """y = act(x @ W.T + b) with ``mpx_linear`` forward; backward = ``mpx_act_backward`` + ``mpx_linear`` on W^T
    (input gradient) + ``mpx_linear_wgrad`` (weight / bias gradients, deterministic split reduction)."""

    @staticmethod
    def forward(ctx, x, weight, bias, act):
        assert x.ndim == 2 and weight.ndim == 2 and x.size(1) == weight.size(1)
        M, K = x.shape
        N = weight.size(0)
        Kp, Np = (K + 3) // 4 * 4, (N + 3) // 4 * 4
        xp = _lib.f32c(x.detach())
        wp = _lib.f32c(weight.detach())
        if Kp != K:  # first layers (4, 67, 259, 7 inputs): zero columns change nothing
            xp = torch.nn.functional.pad(xp, (0, Kp - K))
            wp = torch.nn.functional.pad(wp, (0, Kp - K))
        y = linear(xp, wp, None if bias is None else _lib.f32c(bias.detach()), act)
        ctx.save_for_backward(xp, wp, y if act else None)
        ctx.meta = (act, M, N, K, Np, Kp, bias is not None)
        return y

    @staticmethod
    def backward(ctx, g):
        xp, wp, y = ctx.saved_tensors
        act, M, N, K, Np, Kp, has_bias = ctx.meta
        g = _lib.f32c(g)
        if act:
            dz = torch.empty_like(g)
            _lib.call("mpx_act_backward", _lib.ptr(g), _lib.ptr(y), g.numel(), act, _lib.ptr(dz))
        else:
            dz = g
        if Np != N:  # the 7-wide output layer
            dz = torch.nn.functional.pad(dz, (0, Np - N))
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            wt = torch.nn.functional.pad(wp, (0, 0, 0, Np - N)).t().contiguous() if Np != N else wp.t().contiguous()
            gx = linear(dz, wt, None, 0)[:, :K]  # [M, Kp] -> [M, K]
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            both = torch.empty(Np * Kp + Np, dtype=torch.float32, device=g.device)  # dw | db: one reduction launch
            dw = both[:Np * Kp].view(Np, Kp)
            db = both[Np * Kp:] if has_bias else None
            nscr = _lib.load().mpx_linear_wgrad_scratch(M, Np, Kp)
            scratch = torch.empty(nscr, dtype=torch.float32, device=g.device)
            _lib.call("mpx_linear_wgrad", _lib.ptr(dz), dz.stride(0), _lib.ptr(xp), xp.stride(0), M, Np, Kp, _lib.ptr(dw),
                      _lib.ptr(db), _lib.ptr(scratch))
            gw = dw[:N, :K]
            gb = db[:N] if has_bias else None
        return gx, gw, gb, None


class _GroupNormLeakyFn(torch.autograd.Function):
    """GroupNorm(groups) + LeakyReLU(0.01) on [M, C]: ``mpx_groupnorm_leaky`` forward, ``mpx_groupnorm_leaky_grad`` back."""

    @staticmethod
    def forward(ctx, x, weight, bias, groups, eps):
        xc, w, b = _lib.f32c(x.detach()), _lib.f32c(weight.detach()), _lib.f32c(bias.detach())
        y = groupnorm_leaky(xc, w, b, groups, eps)
        ctx.save_for_backward(xc, w, b)
        ctx.meta = (groups, eps)
        return y

    @staticmethod
    def backward(ctx, g):
        xc, w, b = ctx.saved_tensors
        groups, eps = ctx.meta
        g = _lib.f32c(g)
        M, C = xc.shape
        dx, dw, db = torch.empty_like(xc), torch.empty_like(w), torch.empty_like(b)
        stats = torch.empty(2 * M * groups, dtype=torch.float32, device=xc.device)
        _lib.call("mpx_groupnorm_leaky_grad", _lib.ptr(xc), _lib.ptr(w), _lib.ptr(b), _lib.ptr(g), M, C, groups, float(eps),
                  _lib.ptr(dx), _lib.ptr(dw), _lib.ptr(db), _lib.ptr(stats))
        return dx, dw, db, None, None


def groupnorm_leaky_train(x: torch.Tensor, norm: nn.GroupNorm) -> torch.Tensor:
    return _GroupNormLeakyFn.apply(x, norm.weight, norm.bias, norm.num_groups, norm.eps)


def linear_train(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], act: int = 0) -> torch.Tensor:
    """Differentiable dense layer on the engine's kernels; ``x`` may have leading batch dimensions."""
    lead = x.shape[:-1]
    y = _LinearFn.apply(x.reshape(-1, x.size(-1)), weight, bias, act)
    return y.reshape(lead + (weight.size(0),))


# ---- training path (row N1): differentiable grouping + max-pool on packed distinct-neighbour rows ------
class _PackRows(torch.autograd.Function):
    """QueryAndGroup(use_xyz=True) without the padding: -> rows [R, 3+C]; gradient flows to ``feat`` only."""

    @staticmethod
    def forward(ctx, feat, xyz, xyz_stride, new_xyz, new_stride, feat_stride, C, idx, cnt, offsets, R, dims):
        B, N, npoint, nsample = dims
        ld = (3 + C + 3) // 4 * 4  # rows padded to 16 bytes (zero columns): the GEMMs take them as they are, no padding copy
        rows = torch.empty((R, ld), dtype=torch.float32, device=idx.device)
        _lib.call("mpx_pack_rows_ld", _lib.ptr(xyz), xyz_stride, _lib.ptr(new_xyz), new_stride, _lib.ptr(feat),
                  feat_stride, C, _lib.ptr(idx), _lib.ptr(cnt), _lib.ptr(offsets), B, N, npoint, nsample, _lib.ptr(rows), ld)
        ctx.save_for_backward(idx, cnt, offsets)
        ctx.meta = (C, dims, tuple(feat.shape), feat_stride)
        return rows

    @staticmethod
    def backward(ctx, g):
        idx, cnt, offsets = ctx.saved_tensors
        C, (B, N, npoint, nsample), shape, feat_stride = ctx.meta
        if not ctx.needs_input_grad[0]:
            return (None,) * 12
        if g.dtype != torch.float32 or g.stride(1) != 1 or g.stride(0) < 3 + C:
            g = _lib.f32c(g)
        gf = torch.zeros(shape, dtype=torch.float32, device=g.device)
        _lib.call("mpx_pack_rows_grad_ld", _lib.ptr(g), g.stride(0), C, _lib.ptr(idx), _lib.ptr(cnt), _lib.ptr(offsets), B, N,
                  npoint, nsample, _lib.ptr(gf), gf.stride(1))
        return (gf,) + (None,) * 11


class _SegmentMax(torch.autograd.Function):
    """Max-pool over each query's rows: y [R,C], offsets [Q+1] -> [Q,C]."""

    @staticmethod
    def forward(ctx, y, offsets, Q):
        y = _lib.f32c(y)
        C = y.size(1)
        out = torch.empty((Q, C), dtype=torch.float32, device=y.device)
        arg = torch.empty((Q, C), dtype=torch.int64, device=y.device)
        _lib.call("mpx_segment_max", _lib.ptr(y), C, _lib.ptr(offsets), Q, _lib.ptr(out), C, _lib.ptr(arg))
        ctx.save_for_backward(arg)
        ctx.meta = (y.size(0), C, Q)
        return out

    @staticmethod
    def backward(ctx, g):
        (arg,) = ctx.saved_tensors
        R, C, Q = ctx.meta
        g = _lib.f32c(g)
        gy = torch.zeros((R, C), dtype=torch.float32, device=g.device)
        _lib.call("mpx_segment_max_grad", _lib.ptr(g), g.stride(0), _lib.ptr(arg), Q, C, _lib.ptr(gy))
        return gy, None, None


def _pad4(t: torch.Tensor, rows: bool = False) -> torch.Tensor:
    """Zero-pad the last (or with ``rows`` the first) dimension of a matrix to a multiple of 4."""
    n = t.size(0 if rows else 1)
    n4 = (n + 3) // 4 * 4
    if n4 == n:
        return t
    return torch.nn.functional.pad(t, (0, 0, 0, n4 - n) if rows else (0, n4 - n))


class _MLPChainFn(torch.autograd.Function):
    """A stack of dense layers ``h <- act_i(h @ W_i.T + b_i)`` as ONE autograd node (row N1).

    Versus one ``_LinearFn`` per layer the backward never materialises ``dY * act'(y)`` in a pass of its own: the
    input-gradient GEMM of layer i+1 applies the elementwise backward of layer i's activation in its epilogue
    (``mpx_linear_dact``), and when the stack feeds a segment max-pool (``offsets`` given: the set-abstraction modules)
    the pool's backward applies the last activation's (``mpx_segment_max_grad_act``) -- the last layer's output rows are
    then not kept at all, only the pooled rows and their arg-max.  Arithmetic per element is that of
    ``mpx_act_backward``; gradients equal the per-layer form bit for bit where that form takes the same GEMM kernel (the
    input-gradient product here is always the 128 x 128 tile kernel; ``linear`` splits K or takes its few-row kernel for
    skinny problems: a different summation order).
    Arguments: x [M, K0], acts (tuple of activation codes), offsets (int64 [Q+1] with offsets[0] = 0 and offsets[Q] = M: the
    segments tile the rows; or None), x3 (bool), then W_0, b_0, W_1,
    b_1, ...  Returns the last layer's rows [M, N_last], or the pooled rows [Q, N_last] with ``offsets``.
    ``x3``: the three GEMMs of every layer with >= 128 outputs and >= 1024 rows run in the split-bf16 arithmetic of the
    ``bf16x3`` mode (``mpx_linear_bf16x3`` / ``_dact`` / ``mpx_linear_wgrad_bf16x3``; fp32 master weights, fp32
    accumulation, fp32 activations in memory) -- the engine's form of the reference's ``precision=16`` (run_training.py:112).
    """

    @staticmethod
    def _use_x3(x3, M, N, K):
        return bool(x3) and M >= 1024 and N >= 128 and K >= 16

    @staticmethod
    def forward(ctx, x, acts, offsets, x3, *wb):
        L = len(acts)
        assert len(wb) == 2 * L and x.ndim == 2
        M, K0 = x.shape
        h = _pad4(_lib.f32c(x.detach()))
        xs, ws, meta = [], [], []
        pooled = arg = None
        for i in range(L):
            w, b = wb[2 * i], wb[2 * i + 1]
            N, K = w.shape
            wp = _pad4(_lib.f32c(w.detach()))
            assert h.size(1) == wp.size(1), "layer widths do not chain"
            bias = None if b is None else _lib.f32c(b.detach())
            if offsets is not None and i == L - 1 and FUSED_POOL_FORWARD:
                # the last layer's GEMM pools in its epilogue: its [M, N] rows (2 GB for the second module at batch 256) are
                # neither written nor read back -- the same values and arg-max rows as the two-step form, bit for bit
                Q = offsets.numel() - 1
                seg = torch.repeat_interleave(torch.arange(Q, dtype=torch.int32, device=h.device), offsets[1:] - offsets[:-1],
                                              output_size=M)
                pooled = torch.empty((Q, N), dtype=torch.float32, device=h.device)
                arg = torch.empty((Q, N), dtype=torch.int64, device=h.device)
                keys = torch.empty((Q, N), dtype=torch.int64, device=h.device)
                if _MLPChainFn._use_x3(x3, M, N, wp.size(1)):
                    _lib.call("mpx_linear_segmax_bf16x3", _lib.ptr(h), h.stride(0), _lib.ptr(split_pairs(wp)), _lib.ptr(bias), M, N,
                              wp.size(1), acts[i], _lib.ptr(seg), Q, _lib.ptr(keys), _lib.ptr(pooled), N, _lib.ptr(arg))
                else:
                    _lib.call("mpx_linear_segmax", _lib.ptr(h), h.stride(0), _lib.ptr(wp), _lib.ptr(bias), M, N, wp.size(1),
                              acts[i], _lib.ptr(seg), Q, _lib.ptr(keys), _lib.ptr(pooled), N, _lib.ptr(arg))
                xs.append(h)
                ws.append(wp)
                meta.append((N, K, b is not None))
                break
            if _MLPChainFn._use_x3(x3, M, N, wp.size(1)):
                y = torch.empty((M, N), dtype=torch.float32, device=h.device)
                _lib.call("mpx_linear_bf16x3", _lib.ptr(h), h.stride(0), _lib.ptr(split_pairs(wp)), _lib.ptr(bias), M, N,
                          wp.size(1), acts[i], _lib.ptr(y), N)
            else:
                y = linear(h, wp, bias, acts[i])
            xs.append(h)
            ws.append(wp)
            meta.append((N, K, b is not None))
            h = _pad4(y) if (i + 1 < L and N % 4) else y
        if offsets is not None:
            if pooled is None:
                Q = offsets.numel() - 1
                C = h.size(1)
                pooled = torch.empty((Q, C), dtype=torch.float32, device=h.device)
                arg = torch.empty((Q, C), dtype=torch.int64, device=h.device)
                _lib.call("mpx_segment_max", _lib.ptr(h), C, _lib.ptr(offsets), Q, _lib.ptr(pooled), C, _lib.ptr(arg))
            ctx.save_for_backward(*xs, *ws, pooled, arg, offsets)  # (the last layer's rows are not needed again)
        else:
            ctx.save_for_backward(*xs, *ws, h)
        ctx.meta = (tuple(acts), tuple(meta), M, K0, offsets is not None, bool(x3))
        # LDS window of mpx_pool_dgrad (a hint: longer segments take more passes): twice the mean segment, 32 .. 128 rows
        ctx.max_rows = 128 if offsets is None else min(128, max(32, -(-2 * M // max(offsets.numel() - 1, 1) // 16) * 16))
        return pooled if offsets is not None else h

    @staticmethod
    def backward(ctx, g):
        acts, meta, M, K0, pooled_out, x3 = ctx.meta
        L = len(acts)
        saved = ctx.saved_tensors
        xs, ws = saved[:L], saved[L:2 * L]
        g = _lib.f32c(g)
        dev = g.device
        N_last = meta[-1][0]
        grads = [None] * (2 * L)
        top = L - 1  # the layers [0, top] go through the dense GEMMs below
        if pooled_out and SPARSE_POOL_BACKWARD:
            # the pool hands each (query, channel) gradient to ONE row: the last layer's two products walk those Q * C
            # non-zeros instead of an [M, C] matrix of zeros (never written, never read: mpx_pool_wgrad / mpx_pool_dgrad)
            pooled, arg, offsets = saved[2 * L], saved[2 * L + 1], saved[2 * L + 2]
            Q, C = pooled.shape
            N, K, has_bias = meta[top]
            Kp = ws[top].size(1)
            assert C == N and ws[top].size(0) == N
            lib = _lib.load()
            if ctx.needs_input_grad[4 + 2 * top] or (has_bias and ctx.needs_input_grad[5 + 2 * top]):
                both = torch.empty(N * Kp + N, dtype=torch.float32, device=dev)  # dw | db: one reduction launch
                scratch = torch.empty(lib.mpx_pool_wgrad_scratch(Q, N, Kp), dtype=torch.float32, device=dev)
                _lib.call("mpx_pool_wgrad", _lib.ptr(g), g.stride(0), _lib.ptr(arg), _lib.ptr(pooled), C, Q, N, acts[top],
                          _lib.ptr(xs[top]), xs[top].stride(0), Kp, _lib.ptr(both), _lib.ptr(both[N * Kp:]),
                          _lib.ptr(scratch))
                grads[2 * top] = both[:N * Kp].view(N, Kp)[:, :K]
                grads[2 * top + 1] = both[N * Kp:] if has_bias else None
            if top > 0 or ctx.needs_input_grad[0]:
                below = acts[top - 1] if top > 0 else 0
                dz = torch.empty((M, Kp), dtype=torch.float32, device=dev)  # (every row is written: the segments tile [0, M))
                _lib.call("mpx_pool_dgrad", _lib.ptr(g), g.stride(0), _lib.ptr(arg), _lib.ptr(pooled), C, _lib.ptr(offsets),
                          Q, N, acts[top], _lib.ptr(ws[top]), Kp, _lib.ptr(xs[top]) if below else None, xs[top].stride(0),
                          below, Kp, ctx.max_rows, _lib.ptr(dz), Kp)
            else:
                dz = None
            top -= 1
        elif pooled_out:
            pooled, arg, offsets = saved[2 * L], saved[2 * L + 1], saved[2 * L + 2]
            Q, C = pooled.shape
            dz = torch.empty((M, C), dtype=torch.float32, device=dev)  # (every row is written: the segments tile [0, M))
            _lib.call("mpx_segment_max_grad_act", _lib.ptr(g), g.stride(0), _lib.ptr(arg), _lib.ptr(pooled), C,
                      _lib.ptr(offsets), Q, C, acts[-1], _lib.ptr(dz))
        else:
            y_last = saved[2 * L]
            if acts[-1]:
                dz = torch.empty_like(g)
                _lib.call("mpx_act_backward", _lib.ptr(g), _lib.ptr(y_last), g.numel(), acts[-1], _lib.ptr(dz))
            else:
                dz = g
        for i in range(top, -1, -1):
            N, K, has_bias = meta[i]
            Np, Kp = (N + 3) // 4 * 4, ws[i].size(1)
            dz = _pad4(dz) if dz.size(1) != Np else dz
            lx3 = _MLPChainFn._use_x3(x3, M, N, Kp)
            if ctx.needs_input_grad[4 + 2 * i] or (has_bias and ctx.needs_input_grad[5 + 2 * i]):
                both = torch.empty(Np * Kp + Np, dtype=torch.float32, device=dev)  # dw | db: one reduction launch
                dw = both[:Np * Kp].view(Np, Kp)
                db = both[Np * Kp:] if has_bias else None
                scratch = torch.empty(_lib.load().mpx_linear_wgrad_scratch(M, Np, Kp), dtype=torch.float32, device=dev)
                _lib.call("mpx_linear_wgrad_bf16x3" if lx3 else "mpx_linear_wgrad", _lib.ptr(dz), dz.stride(0), _lib.ptr(xs[i]),
                          xs[i].stride(0), M, Np, Kp, _lib.ptr(dw), _lib.ptr(db), _lib.ptr(scratch))
                grads[2 * i] = dw[:N, :K]
                grads[2 * i + 1] = db[:N] if has_bias else None
            if i > 0 or ctx.needs_input_grad[0]:
                wt = _pad4(ws[i], rows=True).t().contiguous()  # [Kp, Np]
                gx = torch.empty((M, Kp), dtype=torch.float32, device=dev)
                below = acts[i - 1] if i > 0 else 0
                # xs[i] IS the output of layer i - 1 (zero-padded columns: their gradient columns are dropped below)
                if not lx3 and _lib.load().mpx_linear_workspace(M, Kp, Np) > 0:
                    # skinny problem (the reference's batch of 10: a handful of 128 x 128 tiles walking K alone): the
                    # split-K GEMM over the whole chip + the elementwise backward beats the fused epilogue on 8 CUs
                    linear(dz, wt, None, 0, out=gx)
                    if below:
                        _lib.call("mpx_act_backward", _lib.ptr(gx), _lib.ptr(xs[i]), gx.numel(), below, _lib.ptr(gx))
                else:
                    _lib.call("mpx_linear_bf16x3_dact" if lx3 else "mpx_linear_dact", _lib.ptr(dz), dz.stride(0),
                              _lib.ptr(split_pairs(wt) if lx3 else wt), M, Kp, Np, _lib.ptr(xs[i]) if below else None,
                              xs[i].stride(0), below, _lib.ptr(gx), Kp)
                dz = gx
            else:
                dz = None
        gx0 = dz[:, :K0] if (dz is not None and ctx.needs_input_grad[0]) else None
        return (gx0, None, None, None) + tuple(grads)


def mlp_chain_train(x: torch.Tensor, layers, acts, offsets: Optional[torch.Tensor] = None,
                    precision: str = "fp32", offsets_checked: bool = False) -> torch.Tensor:
    """Differentiable stack of dense layers on the engine's kernels (one autograd node, see ``_MLPChainFn``).
    ``layers``: sequence of (weight [N,K], bias or None); ``acts``: one activation code per layer; ``x`` may have leading
    batch dimensions (flattened; not with ``offsets``); ``precision``: "fp32" or "bf16x3" (the large GEMMs of forward and
    backward in split bf16, see ``_MLPChainFn``).  ``offsets`` (int64 [Q+1]) must tile the rows: offsets[0] = 0, ascending,
    offsets[Q] = number of rows -- the pool's backward writes whole segments into an uninitialised buffer, so a gap would
    leak garbage into the gradients; checked here (one host sync) unless the caller vouches with ``offsets_checked``."""
    assert precision in PRECISIONS
    if offsets is not None and not offsets_checked:
        assert x.ndim == 2 and offsets.ndim == 1 and offsets.numel() >= 2, "offsets: int64 [Q+1] over 2-D rows"
        ends = offsets[[0, -1]].tolist()
        assert ends == [0, x.size(0)] and bool((offsets[1:] >= offsets[:-1]).all()), \
            f"offsets must tile the {x.size(0)} rows (got [{ends[0]} .. {ends[1]}], ascending required)"
    lead = x.shape[:-1]
    wb = []
    for w, b in layers:
        wb += [w, b]
    y = _MLPChainFn.apply(x.reshape(-1, x.size(-1)), tuple(int(a) for a in acts), offsets, precision == "bf16x3", *wb)
    return y if offsets is not None else y.reshape(lead + (y.size(-1),))


def segment_offsets(cnt: torch.Tensor) -> torch.Tensor:
    """int64 [Q+1]: start of every query's rows in the packed matrix (a query without a hit keeps one row)."""
    offsets = torch.zeros(cnt.numel() + 1, dtype=torch.int64, device=cnt.device)
    torch.cumsum(cnt.reshape(-1).clamp(min=1), 0, out=offsets[1:])
    return offsets


def sa_module_train(convs: List[nn.Conv2d], xyz: torch.Tensor, xyz_stride: int, new_xyz: torch.Tensor,
                    new_stride: int, feat: torch.Tensor, feat_stride: int, C: int, idx: torch.Tensor,
                    cnt: torch.Tensor, dims: Tuple[int, int, int, int], precision: str = "fp32",
                    offsets: Optional[torch.Tensor] = None, R: Optional[int] = None) -> torch.Tensor:
    """Differentiable set-abstraction MLP + max-pool -> [B, npoint, C_out].  ``feat`` is any tensor whose storage
    holds the point-major features (``feat_stride`` floats between points); it receives the gradient.
    ``offsets`` / ``R`` (``segment_offsets(cnt)`` and its last entry as a host int): a caller that has them already --
    the model reads both modules' row counts with ONE host sync -- passes them in."""
    B, N, npoint, nsample = dims
    if offsets is None:
        offsets = segment_offsets(cnt)
    if R is None:
        R = int(offsets[-1].item())  # a host sync: the row count sizes the activations
    h = _PackRows.apply(feat, xyz, xyz_stride, new_xyz, new_stride, feat_stride, C, idx, cnt, offsets, R, dims)
    layers = [(conv.weight.view(conv.out_channels, -1), conv.bias) for conv in convs]
    # (offsets = cumsum of the clamped counts, R = its last entry = the rows _PackRows made: the segments tile them)
    return mlp_chain_train(h, layers, [1] * len(layers), offsets=offsets, precision=precision,
                           offsets_checked=True).view(B, npoint, -1)


class PointnetSAModule(nn.Module):
    """``PointnetSAModule(npoint=None, radius=None, nsample=None, mlp=[...], bn=False, use_xyz=True)``.

    ``forward(xyz [B,N,3], features [B,C,N]) -> (new_xyz [B,npoint,3] | None, [B,mlp[-1],npoint])``.
    """

    def __init__(self, *, mlp: List[int], npoint: Optional[int] = None, radius: Optional[float] = None,
                 nsample: Optional[int] = None, bn: bool = False, use_xyz: bool = True, precision: str = "fp32"):
        super().__init__()
        assert precision in PRECISIONS
        self.precision = precision  # "fp32" (exact) | "bf16x3" (split-bf16 matrix cores, ~3e-7 on the policy output)
        # skip neighbourhood tiles that hold only ball-query padding (bit-identical output, see launch_sa)
        self.elide_padding = True
        # evaluate the first layer per point / per query where the kernel supports it (fp32, SA2 shape)
        self.factored = True
        if bn:
            raise NotImplementedError("bn=True is not used by the reference (model.py:366-383)")
        assert use_xyz, "the reference relies on use_xyz=True (3 extra input channels)"
        self.npoint, self.radius, self.nsample = npoint, radius, nsample
        spec = list(mlp)
        spec[0] += 3
        layers: List[nn.Module] = []
        for i in range(len(spec) - 1):
            layers.append(nn.Conv2d(spec[i], spec[i + 1], kernel_size=1, bias=True))
            layers.append(nn.ReLU(inplace=True))
        # same container shape as pointnet2_ops: self.mlps = ModuleList([Sequential(conv, relu, ...)])
        self.mlps = nn.ModuleList([nn.Sequential(*layers)])
        self._packed = SAWeights()

    def convs(self) -> List[nn.Conv2d]:
        return [m for m in self.mlps[0] if isinstance(m, nn.Conv2d)]

    def forward(self, xyz: torch.Tensor, features: Optional[torch.Tensor] = None):
        if not xyz.is_cuda:
            raise _lib.MpxError("CPU tensors not supported (reference: model.py:417)")
        xyz = _lib.f32c(xyz)
        B, N, _ = xyz.shape
        convs = self.convs()
        if self.npoint is not None:
            assert features is not None
            C = features.size(1)
            feat_pm = _lib.f32c(features).transpose(1, 2).contiguous()  # [B,N,C] point-major
            idx, new_xyz = furthest_point_sample(xyz, self.npoint, return_xyz=True)
            nbr, cnt = ball_query(self.radius, self.nsample, xyz, new_xyz, return_counts=True)
            if self.training and torch.is_grad_enabled():
                fpm = features.transpose(1, 2).contiguous() if features.requires_grad else feat_pm
                out = sa_module_train(convs, xyz, 3, new_xyz, 3, fpm, C, C, nbr, cnt, (B, N, self.npoint, self.nsample))
                return new_xyz, out.transpose(1, 2).contiguous()
            if use_factored(self, C, convs):
                full = cnt if self.elide_padding else torch.full_like(cnt, self.nsample)
                rows = torch.cat((feat_pm, xyz, torch.zeros_like(xyz[:, :, :1])), dim=2).view(B * N, C + 4)
                ctr_rows = torch.nn.functional.pad(new_xyz, (0, 1)).view(B * self.npoint, 4)
                out = torch.empty((B, self.npoint, convs[-1].out_channels), dtype=torch.float32, device=xyz.device)
                if not hasattr(self, "_split"):
                    self._split = SplitWeights()
                sa_mlp_factored(rows, ctr_rows, nbr, full, self._packed, convs, C, N, _lib.ptr(out), out.stride(1),
                                precision=self.precision, split=self._split)
                return new_xyz, out.transpose(1, 2).contiguous()
            wpack = self._packed.get(convs, C, self.precision)
            out = sa_mlp_fused(xyz, new_xyz, feat_pm, C, C, nbr, wpack, tuple(c.out_channels for c in convs),
                               precision=self.precision, cnt=cnt if self.elide_padding else None)
            return new_xyz, out.transpose(1, 2).contiguous()
        # group-all: one "neighbourhood" holding every point, xyz NOT re-centred
        if self.training and torch.is_grad_enabled():
            h = torch.cat([xyz] + ([features.transpose(1, 2)] if features is not None else []), dim=2)
            for conv in convs:
                h = linear_train(h, conv.weight.view(conv.out_channels, -1), conv.bias, 1)
            return None, h.max(dim=1).values.unsqueeze(-1)
        parts = [xyz]
        if features is not None:
            parts.append(_lib.f32c(features).transpose(1, 2))
        x = torch.cat(parts, dim=2)  # [B,N,3+C]
        K = x.size(2)
        Kp = (K + 3) // 4 * 4
        if Kp != K:
            x = torch.nn.functional.pad(x, (0, Kp - K))
        h = x.reshape(B * N, Kp).contiguous()
        for conv in convs:
            w = conv.weight.detach().reshape(conv.out_channels, -1)
            if w.size(1) != h.size(1):
                w = torch.nn.functional.pad(w, (0, h.size(1) - w.size(1)))
            h = linear(h, _lib.f32c(w), _lib.f32c(conv.bias.detach()), act=1)
        pooled = torch.empty((B, h.size(1)), dtype=torch.float32, device=xyz.device)
        _lib.call("mpx_rowmax", _lib.ptr(h), h.stride(0), B, N, h.size(1), _lib.ptr(pooled), pooled.stride(0))
        return None, pooled.unsqueeze(-1)
