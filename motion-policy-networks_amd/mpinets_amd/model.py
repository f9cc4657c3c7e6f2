"""Policy network and rollouts with the reference's interface (``mpinets/model.py``).

``MotionPolicyNetwork().forward(xyz [B,N,4], q [B,7]) -> [B,7]`` (model.py:75-91), submodule names
and ``state_dict`` keys identical to the reference (``point_cloud_encoder.SA_modules.{0,1,2}.mlps.0.
{0,2,4}``, ``point_cloud_encoder.fc_layer.{0,1,3,4,6}``, ``feature_encoder.{0,2,4,6,8}``,
``decoder.{0,2,4,6}``) so a Lightning checkpoint's ``state_dict`` loads unchanged.
``TrainingMotionPolicyNetwork.rollout(batch, rollout_length, sampler, unnormalize)`` mirrors
model.py:128-183 including the in-place ``xyz[:, :P, :3] = samples`` update.

The forward pass never calls torch compute: FPS / ball query / fused grouped MLP / GEMMs /
GroupNorm all run in ``libmpinets_hip.so`` (fp32 end to end, fp32 MFMA for every contraction).
``pytorch_lightning`` is not required; the classes are plain ``nn.Module``s.
"""
from __future__ import annotations

import ctypes
import os

from typing import Callable, Dict, List, Optional

import torch
from torch import nn

from . import _lib
from .pointnet2 import (PRECISIONS, PointnetSAModule, SplitWeights, groupnorm_leaky, groupnorm_leaky_train,
                        launch_sa, linear, linear_train, linear_x3, mlp_chain_train, sa_mlp_factored, use_factored)
from .utils import unnormalize_franka_joints

ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2


# A handful of problems cannot fill the chip with any single kernel (FPS is ONE workgroup per problem), so below
# this batch size independent branches of the forward are issued on a second HIP stream: the joint-angle encoder
# next to the point-cloud encoder, SA2's sampling + ball query next to SA1's ball query + grouped MLP.
OVERLAP_MAX_BATCH = 512  # (= csrc/policy.hip OVERLAP_MAX_BATCH: the single-call forward makes the same choice)
# From this batch size on the group-all module runs as ONE kernel (mpx_sa3_chain: a workgroup per problem, activations in
# LDS, nothing between the input rows and the pooled row in HBM); below it the layer-by-layer GEMMs (which split K over
# the chip for a handful of problems) are quicker.  (= csrc/policy.hip SA3_CHAIN_MIN_BATCH)
SA3_CHAIN_MIN_BATCH = 256
_SIDE_STREAMS = {}


def side_stream(device) -> "torch.cuda.Stream":
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return st


class MPiNetsPointNet(nn.Module):
    """PointNet++ encoder of the reference (model.py:355-426)."""

    def __init__(self):
        super().__init__()
        self._build_model()

    def _build_model(self):
        self.SA_modules = nn.ModuleList()
        self.SA_modules.append(PointnetSAModule(npoint=512, radius=0.05, nsample=128, mlp=[1, 64, 64, 64], bn=False))
        self.SA_modules.append(PointnetSAModule(npoint=128, radius=0.3, nsample=128, mlp=[64, 128, 128, 256], bn=False))
        self.SA_modules.append(PointnetSAModule(mlp=[256, 512, 512, 1024], bn=False))
        self.fc_layer = nn.Sequential(
            nn.Linear(1024, 4096),
            nn.GroupNorm(16, 4096),
            nn.LeakyReLU(inplace=True),
            nn.Linear(4096, 2048),
            nn.GroupNorm(16, 2048),
            nn.LeakyReLU(inplace=True),
            nn.Linear(2048, 2048),
        )
        self._sa3_w0 = None  # first group-all layer with K padded 259 -> 272 (whole 16-float slabs: direct-to-LDS GEMM)
        self._sa3_pk = None  # (key, mpx_sa3_pack_weights of the group-all module)
        self._sa3_fp = None  # (key, mpx_sa3_front_bf16x3_pack, last layer's permuted weight pairs): bf16x3 mode
        # bf16x3 only: the group-all module's first two layers as one kernel (default) or layer by layer (the round-5 form,
        # bit-identical to the chain through fp32 rows; kept for the tests that pin the fused kernel to it)
        self.sa3_front_fused = True
        self.dense_precision = "fp32"  # "bf16x3": the large dense layers on the bf16 matrix cores (set_precision)
        self.train_precision = "fp32"  # "bf16x3": the grouped / group-all MLPs' training GEMMs in split bf16 (set_training_precision)
        self._split = SplitWeights()
        # bf16x3 only: keep the group-all MLP's activations in the split "pairs" form between layers (default) or as
        # fp32 rows that every layer splits again on its way in -- bit-identical results (tests), pairs are faster
        self.dense_through_pairs = True
        # environments per pass of the encoder (None: the whole batch at once); see forward().  Default: batches up to the
        # bench's 8192 per GPU run in one pass (slabs of 4096 / 2048 measured at 8192: 8.9 -> 5.0 / 3.1 GB peak, +1.8 % /
        # +5.5 % time -- every launch has a tail); larger ones (the whole 65 536-environment configuration) in slabs of <= 8192.
        # Slabs stay above 1024 environments for chunk >= 2048, i.e. inside the dense layers' large-batch launch shape.
        self.workspace_chunk = 8192

    def _lin(self, x, weight, bias, act=0, out=None, source=None):
        if self.dense_precision == "bf16x3":
            return linear_x3(x, weight, bias, act, self._split, out=out, source=source)
        return linear(x, weight, bias, act, out=out)

    def _sa3_through_pairs(self, h: torch.Tensor, c3, B: int, pooled_pairs: bool = False) -> torch.Tensor:
        """The group-all MLP in ``bf16x3`` with its intermediate activations kept in the kernels' pairs form (hi / lo
        bf16 per 16 k-values: the operand of the next layer, staged by DMA and split once, by the epilogue that makes
        them) instead of fp32 rows.  Bit-identical to the fp32-row chain (same split, same accumulation order).  Rows
        go in chunks that keep an operand under the 4 GB a buffer descriptor spans."""
        lib, dev = _lib, h.device
        front = self._sa3_front_pack(h.size(1)) if self.sa3_front_fused else None
        if front is not None:
            # layers 1-2 as ONE kernel (rows divided among the waves, activations in registers, the weights through an LDS
            # ring once per environment: csrc/sa3_front_bf16.hip); its rows carry their k-steps in the kernel's channel
            # order, the last layer's weight pairs have their columns permuted to match.  Equal to the layer-by-layer form
            # to rounding (1e-7 relative), not bit for bit.
            pack, w3p = front
            n2, n3 = c3[1].out_channels, c3[2].out_channels
            pooled = (torch.empty((B, 2 * n3), dtype=torch.bfloat16, device=dev) if pooled_pairs else
                      torch.empty((B, n3), dtype=torch.float32, device=dev))
            step = max(1, min(65535, ((1 << 32) - 4096) // (128 * 4 * n2)))  # environments per call (4 GB descriptors)
            p2 = torch.empty((min(step, B) * 128, 2 * n2), dtype=torch.bfloat16, device=dev)
            for b0 in range(0, B, step):
                nb = min(step, B - b0)
                lib.call("mpx_sa3_front_bf16x3", lib.ptr(h[b0 * 128:]), h.stride(0), nb, 128, lib.ptr(pack), lib.ptr(p2), 2 * n2)
                lib.call("mpx_linear_rowmax_bf16x3_pairs", lib.ptr(p2), 2 * n2, lib.ptr(w3p), lib.ptr(c3[2].bias), nb * 128, n3, n2,
                         128, None if pooled_pairs else lib.ptr(pooled[b0:]), 0 if pooled_pairs else pooled.stride(0),
                         lib.ptr(pooled[b0:]) if pooled_pairs else None, pooled.stride(0) if pooled_pairs else 0)
            return pooled
        w = [self._sa3_first_weight(), c3[1].weight.view(c3[1].out_channels, -1), c3[2].weight.view(c3[2].out_channels, -1)]
        wp = [self._split.get(w[0], c3[0].weight), self._split.get(w[1]), self._split.get(w[2])]
        n1, n2, n3 = (x.size(0) for x in w)
        pooled = (torch.empty((B, 2 * n3), dtype=torch.bfloat16, device=dev) if pooled_pairs else
                  torch.empty((B, n3), dtype=torch.float32, device=dev))
        step = max(1, min(65535, ((1 << 32) - 4096) // (128 * 4 * max(n1, n2))))  # environments per call
        nb0 = min(step, B)
        p1 = torch.empty((nb0 * 128, 2 * n1), dtype=torch.bfloat16, device=dev)
        p2 = torch.empty((nb0 * 128, 2 * n2), dtype=torch.bfloat16, device=dev)
        for b0 in range(0, B, step):
            nb = min(step, B - b0)
            M = nb * 128
            x = h[b0 * 128:]
            lib.call("mpx_linear_bf16x3_to_pairs", lib.ptr(x), h.stride(0), lib.ptr(wp[0]), lib.ptr(c3[0].bias), M, n1,
                     w[0].size(1), ACT_RELU, lib.ptr(p1), 2 * n1)
            lib.call("mpx_linear_bf16x3_pairs", lib.ptr(p1), 2 * n1, lib.ptr(wp[1]), lib.ptr(c3[1].bias), M, n2, n1, ACT_RELU,
                     None, 0, lib.ptr(p2), 2 * n2)
            if pooled_pairs:
                lib.call("mpx_linear_rowmax_bf16x3_pairs", lib.ptr(p2), 2 * n2, lib.ptr(wp[2]), lib.ptr(c3[2].bias), M, n3,
                         n2, 128, None, 0, lib.ptr(pooled[b0:]), pooled.stride(0))
            else:
                lib.call("mpx_linear_rowmax_bf16x3_pairs", lib.ptr(p2), 2 * n2, lib.ptr(wp[2]), lib.ptr(c3[2].bias), M, n3,
                         n2, 128, lib.ptr(pooled[b0:]), pooled.stride(0), None, 0)
        return pooled

    @staticmethod
    def _break_up_pc(pc: torch.Tensor):
        xyz = pc[..., 0:3].contiguous()
        features = pc[..., 3:].transpose(1, 2).contiguous()
        return xyz, features

    def forward_modules(self, point_cloud: torch.Tensor) -> torch.Tensor:
        """Module-by-module evaluation exactly in the reference's shape conventions
        (model.py:409-426); used by tests and by callers that hold their own SA modules."""
        assert point_cloud.size(2) == 4
        xyz, features = self._break_up_pc(point_cloud)
        for module in self.SA_modules:
            xyz, features = module(xyz, features)
        return self._fc(features.squeeze(-1))

    def _fc_through_pairs(self, xp: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The fc head in ``bf16x3`` on an input already in the pairs form [B, 2 * 1024]: every Linear reads pairs (its
        operands go to LDS by DMA), GroupNorm + LeakyReLU writes the next layer's pairs.  Bit-identical to ``_fc``."""
        lib, fc = _lib, self.fc_layer
        B, dev = xp.size(0), xp.device
        h = xp
        for li, gi in ((0, 1), (3, 4), (6, None)):
            lin = fc[li]
            N, K = lin.weight.shape
            last = gi is None
            y = out if (last and out is not None) else torch.empty((B, N), dtype=torch.float32, device=dev)
            lib.call("mpx_linear_bf16x3_pairs", lib.ptr(h), h.stride(0), lib.ptr(self._split.get(lin.weight)), lib.ptr(lin.bias),
                     B, N, K, ACT_NONE, lib.ptr(y), y.stride(0), None, 0)
            if last:
                return y
            gn = fc[gi]
            h = torch.empty((B, 2 * N), dtype=torch.bfloat16, device=dev)
            lib.call("mpx_groupnorm_leaky_to_pairs", lib.ptr(y), lib.ptr(gn.weight), lib.ptr(gn.bias), B, N, gn.num_groups,
                     float(gn.eps), lib.ptr(h), 2 * N)

    def _fc(self, x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        fc = self.fc_layer
        h = self._lin(x, fc[0].weight, fc[0].bias)
        h = groupnorm_leaky(h, fc[1].weight, fc[1].bias, fc[1].num_groups, fc[1].eps, out=h)
        h = self._lin(h, fc[3].weight, fc[3].bias)
        h = groupnorm_leaky(h, fc[4].weight, fc[4].bias, fc[4].num_groups, fc[4].eps, out=h)
        return self._lin(h, fc[6].weight, fc[6].bias, out=out)

    def forward_train(self, point_cloud: torch.Tensor, aux: Optional[dict] = None) -> torch.Tensor:
        """Differentiable forward (training_step, model.py:185-240): sampling / neighbour search / grouping /
        max-pool are this engine's kernels (no padding rows), the dense layers are torch ops under autograd."""
        from .pointnet2 import sa_module_train, segment_offsets

        pc = _lib.f32c(point_cloud)
        B, N, _ = pc.shape
        dev = pc.device
        sa1, sa2, sa3 = self.SA_modules
        lib = _lib
        idx1 = torch.empty((B, sa1.npoint), dtype=torch.int32, device=dev)
        xyz1 = torch.empty((B, sa1.npoint, 3), dtype=torch.float32, device=dev)
        lib.call("mpx_fps", lib.ptr(pc), B, N, 4, sa1.npoint, lib.ptr(idx1), lib.ptr(xyz1), 3)
        nbr1 = torch.empty((B, sa1.npoint, sa1.nsample), dtype=torch.int32, device=dev)
        cnt1 = torch.empty((B, sa1.npoint), dtype=torch.int32, device=dev)
        lib.call("mpx_ball_query", lib.ptr(xyz1), 3, lib.ptr(pc), 4, B, N, sa1.npoint, float(sa1.radius),
                 sa1.nsample, lib.ptr(nbr1), lib.ptr(cnt1))
        # sampling and neighbour search of BOTH modules first: they depend on coordinates only, so the two row counts that
        # size the packed activations reach the host with one sync, after which nothing in the step waits for the device
        idx2 = torch.empty((B, sa2.npoint), dtype=torch.int32, device=dev)
        xyz2 = torch.empty((B, sa2.npoint, 3), dtype=torch.float32, device=dev)
        lib.call("mpx_fps", lib.ptr(xyz1), B, sa1.npoint, 3, sa2.npoint, lib.ptr(idx2), lib.ptr(xyz2), 3)
        nbr2 = torch.empty((B, sa2.npoint, sa2.nsample), dtype=torch.int32, device=dev)
        cnt2 = torch.empty((B, sa2.npoint), dtype=torch.int32, device=dev)
        lib.call("mpx_ball_query", lib.ptr(xyz2), 3, lib.ptr(xyz1), 3, B, sa1.npoint, sa2.npoint, float(sa2.radius),
                 sa2.nsample, lib.ptr(nbr2), lib.ptr(cnt2))
        off1, off2 = segment_offsets(cnt1), segment_offsets(cnt2)
        R1, R2 = (int(v) for v in torch.stack((off1[-1], off2[-1])).tolist())
        # the slab is read in place: coordinates at stride 4, label column = the one input feature (no gradient)
        tp = self.train_precision
        f1 = sa_module_train(sa1.convs(), pc, 4, xyz1, 3, pc[:, :, 3:], 4, 1, nbr1, cnt1, (B, N, sa1.npoint, sa1.nsample), tp,
                             offsets=off1, R=R1)
        f1 = f1.contiguous()
        f2 = sa_module_train(sa2.convs(), xyz1, 3, xyz2, 3, f1, f1.size(2), f1.size(2), nbr2, cnt2,
                             (B, sa1.npoint, sa2.npoint, sa2.nsample), tp, offsets=off2, R=R2)
        h = torch.cat((xyz2, f2), dim=2)  # group-all: absolute coordinates | features
        # (one segment per environment: the pool and its backward are the grouped modules' kernels)
        seg = torch.arange(B + 1, dtype=torch.int64, device=dev) * sa2.npoint
        pooled = mlp_chain_train(h.view(B * sa2.npoint, -1), [(c.weight.view(c.out_channels, -1), c.bias) for c in sa3.convs()],
                                 [ACT_RELU] * 3, offsets=seg, precision=tp, offsets_checked=True)
        self.last_counts = (cnt1, cnt2)
        if aux is not None:
            aux.update(fps_idx1=idx1, xyz1=xyz1, ball_idx1=nbr1, ball_cnt1=cnt1, f1=f1, fps_idx2=idx2, ball_idx2=nbr2,
                       ball_cnt2=cnt2, f3=pooled)
        fc = self.fc_layer  # Linear -> GroupNorm -> LeakyReLU (x2) -> Linear
        h = groupnorm_leaky_train(linear_train(pooled, fc[0].weight, fc[0].bias), fc[1])
        h = groupnorm_leaky_train(linear_train(h, fc[3].weight, fc[3].bias), fc[4])
        return linear_train(h, fc[6].weight, fc[6].bias)

    def _sa3_first_weight(self) -> torch.Tensor:
        conv = self.SA_modules[2].convs()[0]
        key = (conv.weight._version, conv.weight.data_ptr())
        if self._sa3_w0 is None or self._sa3_w0[0] != key:
            w = conv.weight.detach().reshape(conv.out_channels, -1)
            self._sa3_w0 = (key, torch.nn.functional.pad(w, (0, (-w.size(1)) % 16)).contiguous())
        return self._sa3_w0[1]

    def _sa3_front_pack(self, K3: int):
        """(weight pack of ``mpx_sa3_front_bf16x3``, the last layer's weight pairs in its channel order), or None when the
        group-all MLP does not have the widths the kernel is built for."""
        c3 = self.SA_modules[2].convs()
        dims = (K3, c3[0].out_channels, c3[1].out_channels)
        n = _lib.load().mpx_sa3_front_bf16x3_pack_size(*dims)
        if n < 0 or c3[2].out_channels % 16:
            return None
        ps = [p for c in c3 for p in (c.weight, c.bias)]
        key = tuple((p._version, p.data_ptr()) for p in ps) + dims
        if self._sa3_fp is None or self._sa3_fp[0] != key:
            dev = c3[0].weight.device
            w = [_lib.f32c(c.weight.detach().view(c.out_channels, -1)) for c in c3]
            b = [_lib.f32c(c.bias.detach()) for c in c3]
            pack = torch.empty(n, dtype=torch.uint8, device=dev)
            _lib.call("mpx_sa3_front_bf16x3_pack", _lib.ptr(w[0]), w[0].size(1), _lib.ptr(b[0]), _lib.ptr(w[1]), _lib.ptr(b[1]), *dims,
                      _lib.ptr(pack))
            w3p = torch.empty((w[2].size(0), 2 * w[2].size(1)), dtype=torch.bfloat16, device=dev)
            _lib.call("mpx_sa3_front_bf16x3_w3_pairs", _lib.ptr(w[2]), w[2].size(0), w[2].size(1), _lib.ptr(w3p))
            self._sa3_fp = (key, pack, w3p)
        return self._sa3_fp[1], self._sa3_fp[2]

    def _sa3_pack(self, K3: int) -> Optional[torch.Tensor]:
        """The group-all module's three layers in the stream order of ``mpx_sa3_chain`` (None: unsupported widths)."""
        c3 = self.SA_modules[2].convs()
        dims = (K3, c3[0].out_channels, c3[1].out_channels, c3[2].out_channels)
        n = _lib.load().mpx_sa3_pack_size(*dims)
        if n < 0:
            return None
        ps = [p for c in c3 for p in (c.weight, c.bias)]
        key = tuple((p._version, p.data_ptr()) for p in ps) + dims
        if self._sa3_pk is None or self._sa3_pk[0] != key:
            pack = torch.empty(n, dtype=torch.float32, device=c3[0].weight.device)
            w = [_lib.f32c(c.weight.detach().view(c.out_channels, -1)) for c in c3]
            b = [_lib.f32c(c.bias.detach()) for c in c3]
            _lib.call("mpx_sa3_pack_weights", _lib.ptr(w[0]), w[0].size(1), _lib.ptr(b[0]), _lib.ptr(w[1]), _lib.ptr(b[1]),
                      _lib.ptr(w[2]), _lib.ptr(b[2]), *dims, _lib.ptr(pack))
            self._sa3_pk = (key, pack)
        return self._sa3_pk[1]

    def forward(self, point_cloud: torch.Tensor, out: Optional[torch.Tensor] = None,
                aux: Optional[dict] = None, side_work: Optional[Callable[[], object]] = None) -> torch.Tensor:
        """point_cloud [B,N,4] (x,y,z,label) on the GPU -> [B,2048].  ``side_work``: launches of an independent
        branch (the policy's joint-angle encoder) that small batches issue on the second stream, AFTER the first
        sampling kernel is in flight so the host never delays it; large batches simply run it first.

        Engine path: the slab is read in place (stride-4 rows, label column = SA1's feature),
        features stay point-major between modules, SA2 writes straight into the group-all
        module's input rows and nothing is transposed.
        """
        if not point_cloud.is_cuda:
            raise _lib.MpxError("CPU tensors not supported (reference: model.py:417)")
        _lib.require_cuda(point_cloud, self.fc_layer[0].weight)  # (also: operands on the current device)
        assert point_cloud.ndim == 3 and point_cloud.size(2) == 4
        if self.training and torch.is_grad_enabled():
            enc = self.forward_train(point_cloud, aux=aux)
            if out is not None:
                raise _lib.MpxError("out= is an inference-path argument")
            return enc
        pc = _lib.f32c(point_cloud)
        B, N, _ = pc.shape
        dev = pc.device
        chunk = self.workspace_chunk
        if chunk and aux is None and B > chunk:
            # Large batches go through the encoder in slabs of <= `workspace_chunk` environments: the intermediates
            # (neighbour rows, per-point first-layer rows, module outputs: 0.73 MB per environment) are slab-sized and
            # reused, so the workspace is 0.73 MB x chunk whatever B is (8192 environments: 3 GB instead of 6; the whole
            # 65 536-environment configuration: the same 3 GB instead of 48).  Environments never interact and every slab
            # is large enough for the launch shapes of the whole batch: bit-identical results (tests/test_gpu_policy.py).
            if side_work is not None:
                side_work()
            if out is None:
                out = torch.empty((B, self.fc_layer[6].out_features), dtype=torch.float32, device=dev)
            counts = []
            n_slabs = -(-B // chunk)
            for i in range(n_slabs):  # near-equal slabs: no small tail with other launch shapes
                b0, b1 = B * i // n_slabs, B * (i + 1) // n_slabs
                self.forward(pc[b0:b1], out=out[b0:b1])
                counts.append(self.last_counts)
            self.last_counts = tuple(torch.cat([c[k] for c in counts]) for k in (0, 1))
            return out
        sa1, sa2, sa3 = self.SA_modules
        lib = _lib
        # ---- SA1 -------------------------------------------------------------------------------
        idx1 = torch.empty((B, sa1.npoint), dtype=torch.int32, device=dev)
        xyz1 = torch.empty((B, sa1.npoint, 3), dtype=torch.float32, device=dev)
        lib.call("mpx_fps", lib.ptr(pc), B, N, 4, sa1.npoint, lib.ptr(idx1), lib.ptr(xyz1), 3)
        nbr1 = torch.empty((B, sa1.npoint, sa1.nsample), dtype=torch.int32, device=dev)
        cnt1 = torch.empty((B, sa1.npoint), dtype=torch.int32, device=dev)
        c1 = sa1.convs()
        w1 = sa1._packed.get(c1, 1, sa1.precision)
        # SA1 output rows carry [f1 (64) | xyz1 (3) | 0]: the operand of SA2's per-point first-layer GEMM
        C1o = c1[-1].out_channels
        f1buf = torch.empty((B, sa1.npoint, C1o + 4), dtype=torch.float32, device=dev)
        f1 = f1buf[:, :, :C1o]
        # SA2's inputs (the group-all rows [xyz2 | f2 | 0] it will write into, its samples and neighbours)
        c2 = sa2.convs()
        C2o = c2[-1].out_channels
        K3 = (3 + C2o + 15) // 16 * 16  # [xyz2 | f2 | 0...]: whole 16-float GEMM slabs
        sa3_in = torch.empty((B, sa2.npoint, K3), dtype=torch.float32, device=dev)
        # xyz2 and f2 are written by the sampling and SA2 kernels: only the padding needs zeros -- and column 3, which
        # the per-query first-layer GEMM reads (times a zero weight) before SA2 has written it
        sa3_in[:, :, 3 + C2o:] = 0
        sa3_in[:, :, 3] = 0
        idx2 = torch.empty((B, sa2.npoint), dtype=torch.int32, device=dev)
        nbr2 = torch.empty((B, sa2.npoint, sa2.nsample), dtype=torch.int32, device=dev)
        cnt2 = torch.empty((B, sa2.npoint), dtype=torch.int32, device=dev)

        # the fused grouped-MLP kernels take the hit counts and never look past them: the ball queries then write the hit
        # slots only (most of a 128-slot row is padding).  With `aux` (callers that want the reference's full index rows)
        # or with padding elision off, the rows are padded like pointnet2_ops pads them.
        bq1 = "mpx_ball_query_hits" if (aux is None and sa1.elide_padding) else "mpx_ball_query"
        bq2 = "mpx_ball_query_hits" if (aux is None and sa2.elide_padding) else "mpx_ball_query"

        def sample_sa2():  # needs only xyz1
            lib.call("mpx_fps", lib.ptr(xyz1), B, sa1.npoint, 3, sa2.npoint, lib.ptr(idx2), lib.ptr(sa3_in), K3)
            lib.call(bq2, lib.ptr(sa3_in), K3, lib.ptr(xyz1), 3, B, sa1.npoint, sa2.npoint,
                     float(sa2.radius), sa2.nsample, lib.ptr(nbr2), lib.ptr(cnt2))

        def module_sa1():
            lib.call(bq1, lib.ptr(xyz1), 3, lib.ptr(pc), 4, B, N, sa1.npoint, float(sa1.radius),
                     sa1.nsample, lib.ptr(nbr1), lib.ptr(cnt1))
            # (SA1 also completes its rows to [f1 | xyz1 | 0] when SA2's first layer is evaluated per point)
            state["centre_done"] = launch_sa(sa1.precision, lib.ptr(pc), 4, lib.ptr(xyz1), 3, lib.ptr(pc) + 12, 4, 1, nbr1,
                                             cnt1 if sa1.elide_padding else None, B, N, sa1.npoint, sa1.nsample, w1,
                                             tuple(c.out_channels for c in c1), lib.ptr(f1), f1.stride(1),
                                             append_centre=want_rows)

        want_rows = use_factored(sa2, C1o, c2)
        state = {"centre_done": False}
        keep = None
        if B <= OVERLAP_MAX_BATCH:  # two independent chains, two streams (buffers were allocated above, on `main`)
            main, side = torch.cuda.current_stream(), side_stream(dev)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                sample_sa2()
                if side_work is not None:
                    keep = side_work()  # (its temporaries stay referenced until `main` has waited for `side`)
            module_sa1()
            main.wait_stream(side)
        else:
            if side_work is not None:
                side_work()
            module_sa1()
            sample_sa2()
        del keep
        if want_rows:
            if not state["centre_done"]:
                lib.call("mpx_append_columns", lib.ptr(xyz1), 3, 3, 1, B * sa1.npoint, lib.ptr(f1buf), C1o + 4, C1o)
            sa_mlp_factored(f1buf.view(B * sa1.npoint, C1o + 4), sa3_in.view(B * sa2.npoint, K3)[:, :4], nbr2,
                            cnt2 if sa2.elide_padding else torch.full_like(cnt2, sa2.nsample), sa2._packed, c2, C1o,
                            sa1.npoint, lib.ptr(sa3_in) + 12, K3, precision=sa2.precision, split=self._split)
        else:
            w2 = sa2._packed.get(c2, C1o, sa2.precision)
            launch_sa(sa2.precision, lib.ptr(xyz1), 3, lib.ptr(sa3_in), K3, lib.ptr(f1), f1.stride(1), C1o, nbr2,
                      cnt2 if sa2.elide_padding else None, B, sa1.npoint, sa2.npoint, sa2.nsample, w2,
                      tuple(c.out_channels for c in c2), lib.ptr(sa3_in) + 12, K3)
        # ---- SA3 (group-all): three GEMMs over B*128 rows + max over each environment's rows ------------
        c3 = sa3.convs()
        h = sa3_in.view(B * sa2.npoint, K3)
        if (self.dense_precision == "bf16x3" and sa2.npoint == 128 and self.dense_through_pairs
                and all(c.out_channels % 16 == 0 for c in c3)):
            self.last_counts = (cnt1, cnt2)
            if aux is None and c3[2].out_channels % 16 == 0:  # (aux wants the pooled features as fp32)
                return self._fc_through_pairs(self._sa3_through_pairs(h, c3, B, pooled_pairs=True), out=out)
            pooled = self._sa3_through_pairs(h, c3, B)
            if aux is not None:
                aux.update(fps_idx1=idx1, xyz1=xyz1, ball_idx1=nbr1, ball_cnt1=cnt1, f1=f1, fps_idx2=idx2, ball_idx2=nbr2,
                           ball_cnt2=cnt2, sa3_in=sa3_in, f3=pooled)
            return self._fc(pooled, out=out)
        pack3 = (self._sa3_pack(K3) if self.dense_precision == "fp32" and sa2.npoint == 128 and B >= SA3_CHAIN_MIN_BATCH
                 else None)
        if pack3 is not None:  # the whole module as one kernel: nothing between the rows and the pooled row in HBM
            pooled = torch.empty((B, c3[2].out_channels), dtype=torch.float32, device=dev)
            lib.call("mpx_sa3_chain", lib.ptr(h), K3, B, 128, lib.ptr(pack3), K3, c3[0].out_channels, c3[1].out_channels,
                     c3[2].out_channels, lib.ptr(pooled), pooled.stride(0))
            self.last_counts = (cnt1, cnt2)
            if aux is not None:
                aux.update(fps_idx1=idx1, xyz1=xyz1, ball_idx1=nbr1, ball_cnt1=cnt1, f1=f1, fps_idx2=idx2, ball_idx2=nbr2,
                           ball_cnt2=cnt2, sa3_in=sa3_in, f3=pooled)
            return self._fc(pooled, out=out)
        h = self._lin(h, self._sa3_first_weight(), c3[0].bias, ACT_RELU, source=c3[0].weight)
        h = self._lin(h, c3[1].weight.view(c3[1].out_channels, -1), c3[1].bias, ACT_RELU)
        # last layer + max over each environment's 128 points in one kernel (nothing [B*128,1024] is stored)
        w3 = c3[2].weight.view(c3[2].out_channels, -1)
        pooled = torch.empty((B, w3.size(0)), dtype=torch.float32, device=dev)
        # (a handful of problems leave the fused kernel 8 workgroups per problem: there the split-K layer followed
        # by the row-max kernel is quicker -- 45 -> 22 us for one problem)
        if sa2.npoint == 128 and (B > 8 or self.dense_precision == "bf16x3"):
            for b0 in range(0, B, 65535):
                nb = min(65535, B - b0)
                if self.dense_precision == "bf16x3":
                    lib.call("mpx_linear_rowmax_bf16x3", lib.ptr(h[b0 * 128:]), h.stride(0), lib.ptr(self._split.get(w3)),
                             lib.ptr(c3[2].bias), nb * 128, w3.size(0), w3.size(1), 128, lib.ptr(pooled[b0:]),
                             pooled.stride(0))
                else:
                    lib.call("mpx_linear_rowmax", lib.ptr(h[b0 * 128:]), h.stride(0), lib.ptr(w3), lib.ptr(c3[2].bias),
                             nb * 128, w3.size(0), w3.size(1), 128, lib.ptr(pooled[b0:]), pooled.stride(0))
        else:
            h = self._lin(h, w3, c3[2].bias, ACT_RELU)
            for b0 in range(0, B, 65535):
                nb = min(65535, B - b0)
                lib.call("mpx_rowmax", lib.ptr(h[b0 * sa2.npoint:]), h.stride(0), nb, sa2.npoint, h.size(1),
                         lib.ptr(pooled[b0:]), pooled.stride(0))
        self.last_counts = (cnt1, cnt2)  # distinct-neighbour counts of the latest forward (bench accounting)
        if aux is not None:
            aux.update(fps_idx1=idx1, xyz1=xyz1, ball_idx1=nbr1, ball_cnt1=cnt1, f1=f1, fps_idx2=idx2, ball_idx2=nbr2,
                       ball_cnt2=cnt2, sa3_in=sa3_in, f3=pooled)
        return self._fc(pooled, out=out)


class MotionPolicyNetwork(nn.Module):
    """The default MPiNets architecture (model.py:35-91)."""

    def __init__(self):
        super().__init__()
        self.point_cloud_encoder = MPiNetsPointNet()
        self.feature_encoder = nn.Sequential(
            nn.Linear(7, 32), nn.LeakyReLU(), nn.Linear(32, 64), nn.LeakyReLU(), nn.Linear(64, 128),
            nn.LeakyReLU(), nn.Linear(128, 128), nn.LeakyReLU(), nn.Linear(128, 64),
        )
        self.decoder = nn.Sequential(
            nn.Linear(2048 + 64, 512), nn.LeakyReLU(), nn.Linear(512, 256), nn.LeakyReLU(), nn.Linear(256, 128),
            nn.LeakyReLU(), nn.Linear(128, 7),
        )
        self._q_w0 = None

    @property
    def device(self):
        return next(self.parameters()).device

    def set_precision(self, precision: str) -> "MotionPolicyNetwork":
        """Arithmetic of the matrix work (the two grouped MLPs and the large dense layers): ``"fp32"`` = exact
        fp32 MFMA (default, the parity path) or ``"bf16x3"`` = split-bf16 on the bf16 matrix cores (each product
        as hi*hi + hi*lo + lo*hi, fp32 accumulate; policy output within ~3e-7 of fp32).  Everything else (FPS,
        ball query, GroupNorm, the small joint-encoder / decoder layers, FK, SDF) is fp32 in both modes."""
        assert precision in PRECISIONS, precision
        for sa in self.point_cloud_encoder.SA_modules:
            sa.precision = precision
        self.point_cloud_encoder.dense_precision = precision
        return self

    def set_training_precision(self, precision: str) -> "MotionPolicyNetwork":
        """Arithmetic of the training GEMMs of the grouped / group-all MLPs (forward, input gradient, weight gradient of the
        layers with >= 128 outputs over >= 1024 rows): ``"fp32"`` (default) or ``"bf16x3"`` -- split bf16 on the bf16 matrix
        cores with fp32 accumulation, fp32 master weights and fp32 activations in memory: this engine's form of the
        reference's mixed-precision training (``precision=16``, run_training.py:112), with gradients within ~1e-4 relative
        of the fp32 path instead of fp16's ~1e-3.  Sampling, grouping, losses, GroupNorm and the optimizer stay fp32."""
        assert precision in PRECISIONS, precision
        self.point_cloud_encoder.train_precision = precision
        return self

    def set_factored(self, on: bool) -> "MotionPolicyNetwork":
        """Evaluate SA2's first layer per point / per query (``mpx_sa_mlp_factored``, default on) or per
        (query, neighbour) row like the reference's op order (``mpx_sa_mlp``); the results agree to ~1e-7."""
        for sa in self.point_cloud_encoder.SA_modules:
            sa.factored = bool(on)
        return self

    def set_elide_padding(self, on: bool) -> "MotionPolicyNetwork":
        """Skip neighbourhood tiles made only of ball-query padding (default on; output bit-identical)."""
        for sa in self.point_cloud_encoder.SA_modules:
            sa.elide_padding = bool(on)
        return self

    def invalidate_caches(self) -> "MotionPolicyNetwork":
        """Drop every derived weight buffer (MFMA-stream packs, bf16 hi/lo planes, padded first layers).  They refresh
        by themselves when a parameter is modified through torch (``_version`` changes: optimizer steps,
        ``load_state_dict``, ``copy_``); writes through ``param.data`` (EMA swaps, manual init) do not bump the
        version -- call this after them."""
        enc = self.point_cloud_encoder
        for sa in enc.SA_modules:
            sa._packed.packs.clear()
            sa._packed._fact = None
            if hasattr(sa, "_split"):
                sa._split.cache.clear()
        enc._split.cache.clear()
        enc._sa3_w0 = None
        enc._sa3_pk = None
        enc._sa3_fp = None
        self._q_w0 = None
        return self

    def configure_optimizers(self):
        return torch.optim.Adam(self.parameters(), lr=1e-4)

    @classmethod
    def load_from_checkpoint(cls, path: str, map_location="cpu", trust_checkpoint: bool = False, **kwargs):
        """Reads a Lightning ``.ckpt`` (``{'state_dict': ...}``) or a bare state dict (run_inference.py:262) with
        ``torch.load(weights_only=True)``: tensors and plain containers only, no code runs.

        Real Lightning checkpoints may also pickle callback / hyper-parameter objects, which the safe loader refuses.
        Unpickling those executes arbitrary code from the file, so it is opt-in: ``trust_checkpoint=True`` (a local file
        you produced yourself) re-reads with ``weights_only=False`` like ``LightningModule.load_from_checkpoint`` does.
        Without it the refusal is raised with that hint; other failures (truncated file, bad device map) are never
        retried."""
        import pickle

        try:
            ckpt = torch.load(path, map_location=map_location, weights_only=True)
        except pickle.UnpicklingError as e:
            if not trust_checkpoint:
                raise pickle.UnpicklingError(
                    f"{path}: the checkpoint pickles objects beyond tensors ({e}).  If the file is yours, pass "
                    "trust_checkpoint=True to load it with weights_only=False (this executes code stored in the file).") from e
            import warnings

            warnings.warn(f"{path}: loading with weights_only=False (trust_checkpoint=True): pickled code in the file runs")
            ckpt = torch.load(path, map_location=map_location, weights_only=False)
        sd = ckpt.get("state_dict", ckpt)
        mdl = cls(**kwargs)
        mdl.load_state_dict({k: v for k, v in sd.items() if not k.startswith("loss_fun")})
        return mdl

    def _q_first_weight(self) -> torch.Tensor:
        lin = self.feature_encoder[0]
        key = (lin.weight._version, lin.weight.data_ptr())
        if self._q_w0 is None or self._q_w0[0] != key:
            self._q_w0 = (key, torch.nn.functional.pad(lin.weight.detach(), (0, 1)).contiguous())
        return self._q_w0[1]

    # ---- the single-call C entry point (mpx_policy_forward): what a caller without Python would use ----------
    class _NativeWeights(ctypes.Structure):  # field order = struct mpx_policy_weights (include/mpinets_hip.h)
        _fields_ = [(n, ctypes.c_void_p) for n in ("sa1_pack", "sa2_pack", "sa2_wpoint", "sa2_wcentre", "sa2_nb1")] + \
                   [(n, ctypes.c_void_p * k) for n, k in (("sa3_w", 3), ("sa3_b", 3), ("fc_w", 3), ("fc_b", 3), ("gn_g", 2),
                                                          ("gn_b", 2), ("qe_w", 5), ("qe_b", 5), ("de_w", 4), ("de_b", 4))] + \
                   [("sa3_pack", ctypes.c_void_p)]

    def native_weights(self):
        """-> (struct mpx_policy_weights, tensors it points to).  Valid until a parameter changes."""
        enc = self.point_cloud_encoder
        sa1, sa2, sa3 = enc.SA_modules
        c1, c2, c3 = sa1.convs(), sa2.convs(), sa3.convs()
        f = lambda t: _lib.f32c(t.detach())
        wp, wc, nb1 = sa2._packed.factored(c2, c1[-1].out_channels)
        lins = lambda seq: [m for m in seq if isinstance(m, nn.Linear)]
        fc, fe, de = lins(enc.fc_layer), lins(self.feature_encoder), lins(self.decoder)
        gns = [m for m in enc.fc_layer if isinstance(m, nn.GroupNorm)]
        groups = {
            "sa3_w": [enc._sa3_first_weight()] + [f(c.weight.view(c.out_channels, -1)) for c in c3[1:]],
            "sa3_b": [f(c.bias) for c in c3],
            "fc_w": [f(m.weight) for m in fc], "fc_b": [f(m.bias) for m in fc],
            "gn_g": [f(m.weight) for m in gns], "gn_b": [f(m.bias) for m in gns],
            "qe_w": [self._q_first_weight()] + [f(m.weight) for m in fe[1:]], "qe_b": [f(m.bias) for m in fe],
            "de_w": [f(m.weight) for m in de], "de_b": [f(m.bias) for m in de],
        }
        single = {"sa1_pack": sa1._packed.get(c1, 1, "fp32"), "sa2_pack": sa2._packed.get(c2, c1[-1].out_channels, "fp32"),
                  "sa2_wpoint": wp, "sa2_wcentre": wc, "sa2_nb1": nb1,
                  "sa3_pack": enc._sa3_pack((3 + c2[-1].out_channels + 15) // 16 * 16)}
        w = self._NativeWeights()
        for k, t in single.items():  # (sa3_pack is None for widths the fused chain does not cover: NULL = layer by layer)
            setattr(w, k, t.data_ptr() if t is not None else None)
        for k, ts in groups.items():
            setattr(w, k, (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts]))
        return w, (single, groups)

    def forward_native(self, xyz: torch.Tensor, q: torch.Tensor) -> torch.Tensor:
        """The same forward through ``mpx_policy_forward`` (fp32): one C call, caller-provided workspace."""
        B, N, _ = xyz.shape
        w, keep = self.native_weights()
        need = _lib.load().mpx_policy_workspace(B, N)
        ws = torch.empty(need, dtype=torch.uint8, device=xyz.device)
        dq = torch.empty((B, 7), dtype=torch.float32, device=xyz.device)
        _lib.call("mpx_policy_forward", ctypes.addressof(w), _lib.ptr(_lib.f32c(xyz)), N, _lib.ptr(_lib.f32c(q)), B,
                  _lib.ptr(dq), _lib.ptr(ws), need)
        del keep
        return dq

    def forward(self, xyz: torch.Tensor, q: torch.Tensor, aux: Optional[dict] = None) -> torch.Tensor:
        """xyz [B,N,4], q [B,7] normalised to [-1,1] -> displacement [B,7] (normalised space)."""
        if not xyz.is_cuda:
            raise _lib.MpxError("CPU tensors not supported (reference: model.py:417)")
        B = xyz.size(0)
        dev = xyz.device
        if self.training and torch.is_grad_enabled():  # differentiable path (training_step)
            pc_encoding = self.point_cloud_encoder(xyz, aux=aux)

            def mlp(seq, x):  # Linear + LeakyReLU stacks: one autograd node, activation backward in the GEMM epilogues
                layers = [m for m in seq if isinstance(m, nn.Linear)]
                return mlp_chain_train(x, [(lin.weight, lin.bias) for lin in layers],
                                       [ACT_LEAKY] * (len(layers) - 1) + [ACT_NONE])

            return mlp(self.decoder, torch.cat((pc_encoding, mlp(self.feature_encoder, _lib.f32c(q))), dim=1))
        cat = torch.empty((B, 2048 + 64), dtype=torch.float32, device=dev)
        # (small batches: beside the point-cloud encoder, on its second stream -- see OVERLAP_MAX_BATCH)
        self.point_cloud_encoder(xyz, out=cat[:, :2048], aux=aux, side_work=lambda: self.encode_configuration(q, out=cat[:, 2048:]))
        if aux is not None:
            aux["encoding"] = cat[:, :2048]
        dq = self.decode(cat)
        return dq

    def encode_configuration(self, q: torch.Tensor, out: Optional[torch.Tensor] = None):
        """``feature_encoder`` (model.py:45-55): q [B,7] -> [B,64], written to ``out`` when given.  Returns the output
        followed by the temporaries (callers on a side stream keep them referenced until the streams have joined)."""
        fe = self.feature_encoder
        B, dev = q.size(0), q.device
        q8 = torch.zeros((B, 8), dtype=torch.float32, device=dev)
        q8[:, :7] = q
        h1 = linear(q8, self._q_first_weight(), fe[0].bias, ACT_LEAKY)
        h2 = linear(h1, fe[2].weight, fe[2].bias, ACT_LEAKY)
        h3 = linear(h2, fe[4].weight, fe[4].bias, ACT_LEAKY)
        h4 = linear(h3, fe[6].weight, fe[6].bias, ACT_LEAKY)
        y = linear(h4, fe[8].weight, fe[8].bias, ACT_NONE, out=out)
        return y, q8, h1, h2, h3, h4

    def decode(self, cat: torch.Tensor) -> torch.Tensor:
        """``decoder`` (model.py:56-64) on ``[pc_encoding (2048) | feature_encoding (64)]`` rows -> [B,7]."""
        de = self.decoder
        h = self.point_cloud_encoder._lin(cat, de[0].weight, de[0].bias, ACT_LEAKY)
        h = linear(h, de[2].weight, de[2].bias, ACT_LEAKY)
        h = linear(h, de[4].weight, de[4].bias, ACT_LEAKY)
        return linear(h, de[6].weight, de[6].bias, ACT_NONE)


class TrainingMotionPolicyNetwork(MotionPolicyNetwork):
    """Adds the training / rollout / validation helpers of the reference (model.py:94-318)."""

    def __init__(self, num_robot_points: int, point_match_loss_weight: float = 1.0,
                 collision_loss_weight: float = 1.0):
        super().__init__()
        from .loss import CollisionAndBCLossContainer

        self.num_robot_points = num_robot_points
        self.point_match_loss_weight = point_match_loss_weight
        self.collision_loss_weight = collision_loss_weight
        self.fk_sampler = None
        self.collision_sampler = None
        self.loss_fun = CollisionAndBCLossContainer()
        self.logged: Dict[str, torch.Tensor] = {}

    def log(self, name: str, value: torch.Tensor) -> None:
        """Stand-in for LightningModule.log: keeps the latest detached value per name."""
        self.logged[name] = value.detach()

    def training_step(self, batch: Dict[str, torch.Tensor], batch_idx: int = 0) -> torch.Tensor:
        """model.py:185-240: one supervised step -> the weighted loss (call ``.backward()`` on it)."""
        xyz, q = batch["xyz"], batch["configuration"]
        y_hat = torch.clamp(q + self(xyz, q), min=-1, max=1)
        collision_loss, point_match_loss = self.loss_fun(
            y_hat, batch["cuboid_centers"], batch["cuboid_dims"], batch["cuboid_quats"], batch["cylinder_centers"],
            batch["cylinder_radii"], batch["cylinder_heights"], batch["cylinder_quats"], batch["supervision"])
        self.log("point_match_loss", point_match_loss)
        self.log("collision_loss", collision_loss)
        val_loss = self.point_match_loss_weight * point_match_loss + self.collision_loss_weight * collision_loss
        self.log("val_loss", val_loss)
        return val_loss

    def rollout(self, batch: Dict[str, torch.Tensor], rollout_length: int,
                sampler: Callable[[torch.Tensor], torch.Tensor], unnormalize: bool = False) -> List[torch.Tensor]:
        """model.py:128-183: ``q = clamp(q + self(xyz, q), -1, 1)``; resample the robot points at the
        new configuration and overwrite ``xyz[:, :P, :3]`` in place; returns rollout_length+1 tensors."""
        xyz, q = batch["xyz"], batch["configuration"]
        if q.ndim == 1:
            xyz = xyz.unsqueeze(0)
            q = q.unsqueeze(0)
        if unnormalize:
            q_unnorm = unnormalize_franka_joints(q)
            assert isinstance(q_unnorm, torch.Tensor)
            trajectory = [q_unnorm]
        else:
            trajectory = [q]
        for _ in range(rollout_length):
            q = torch.clamp(q + self(xyz, q), min=-1, max=1)
            q_unnorm = unnormalize_franka_joints(q).type_as(q)
            trajectory.append(q_unnorm if unnormalize else q)
            samples = sampler(q_unnorm).type_as(xyz)
            xyz[:, : samples.shape[1], :3] = samples
        return trajectory

    def sample(self, q: torch.Tensor) -> torch.Tensor:
        assert self.fk_sampler is not None
        return self.fk_sampler.sample(q, self.num_robot_points)

    @torch.no_grad()
    def validation_step(self, batch: Dict[str, torch.Tensor], batch_idx: int = 0, rollout_length: int = 69):
        """model.py:252-318: rollout, final target error, swept-sphere collision rate."""
        from .geometry import TorchCuboids, TorchCylinders
        from .robot import FrankaCollisionSampler, FrankaSampler

        dev = batch["xyz"].device
        if self.fk_sampler is None:
            self.fk_sampler = FrankaSampler(dev, use_cache=True)
        if self.collision_sampler is None:
            self.collision_sampler = FrankaCollisionSampler(dev, with_base_link=False)
        rollout = self.rollout(batch, rollout_length, self.sample, unnormalize=True)
        eff = self.fk_sampler.end_effector_pose(rollout[-1])
        position_error = torch.linalg.vector_norm(eff[:, :3, -1] - batch["target_position"], dim=1)
        cuboids = TorchCuboids(batch["cuboid_centers"], batch["cuboid_dims"], batch["cuboid_quats"])
        cylinders = TorchCylinders(batch["cylinder_centers"], batch["cylinder_radii"], batch["cylinder_heights"],
                                   batch["cylinder_quats"])
        traj = torch.stack(rollout, dim=1)  # [B, L+1, 7]
        has_collision = self.collision_sampler.check(traj, cuboids, cylinders)
        B = traj.size(0)
        return {"avg_target_error": torch.mean(position_error),
                "avg_collision_rate": torch.count_nonzero(has_collision) / B}

    def validation_step_end(self, batch_parts: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """model.py:320-334 (Lightning's per-step aggregation over devices): the mean of each statistic's parts."""
        return {"avg_target_error": torch.mean(batch_parts["avg_target_error"]),
                "avg_collision_rate": torch.mean(batch_parts["avg_collision_rate"])}

    def validation_epoch_end(self, validation_step_outputs) -> None:
        """model.py:336-352: epoch means of the two statistics, through ``log`` (here: kept in ``self.logged``)."""
        self.log("avg_target_error", torch.mean(torch.stack([x["avg_target_error"] for x in validation_step_outputs])))
        self.log("avg_collision_rate", torch.mean(torch.stack([x["avg_collision_rate"] for x in validation_step_outputs])))
