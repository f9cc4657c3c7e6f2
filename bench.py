#!/usr/bin/env python
"""bench.py -- env-steps/s of the closed-loop policy step (FK + SDF collision + PointNet++).

    python bench.py --gpus N --steps K --warmup W        (N > 1: starts its own N ranks, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Launched plainly with --gpus N > 1 it re-executes itself under torch.distributed.run (like the reference's launcher
starts its own ranks, run_training.py:71-77,106-115) and fails loudly when fewer than N GPUs are visible
(development only: MPX_SHARE_GPU=1 lets the ranks share the visible GPUs over gloo; the JSON then says so).

Workload (BASELINE.json metric "at 8192 envs" = the per-GPU share of its largest configuration, configs[4]:
"mixed tabletop/cubby/dresser 65536 envs, closed-loop point-cloud re-render + policy step, 8 GPUs"): every
rank owns 8192 independent synthetic planning problems, one third each tabletop / cubby / dresser-like scenes
(weak scaling; no collective on the step, one final gather).  One STEP = one pass of the hot path over the
whole batch with inputs resident in HBM:
scene cloud re-rendered from the primitives (4096 points per environment) -> policy forward (FPS 6272->512,
ball query, fused grouped MLP, FPS 512->128, ball query, fused MLP, group-all MLP, heads) -> joint update ->
FK + robot-cloud refresh in place -> swept-sphere SDF collision check of the new configuration.
fp32 end to end (fp32 MFMA for every contraction).

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (the SA2 fused grouped MLP),
timed live with HIP events on the launch stream inside the timed region; `cpu_baseline` is the
oracle (a port, not the product) timed on this host's cores on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "motion-policy-networks_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np
import torch

SA1_FLOPS = 65536 * 8448 * 2  # 512*128 rows x (4*64 + 64*64 + 64*64) MACs     
SA2_FLOPS = 16384 * 57728 * 2  # 128*128 rows x (67*128 + 128*128 + 128*256) MACs
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak


def cpu_step(orc, ft, sd, xyz, qn, prim, tables, torch_threads=0):
    """One pass of the hot path over a batch on the host, by the oracle (a port, not the product).  The policy forward
    is the oracle's torch restatement in fp32 on `torch_threads` threads (the identical op sequence on
    all host cores), or its float64-accumulate numpy form when `torch_threads` is 0."""
    (c, r, l), (tp, tl) = tables
    if torch_threads:
        torch.set_num_threads(torch_threads)
        sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
        with torch.no_grad():  # (chunks of 32 environments: the grouped activations of one chunk are ~2 GB)
            dq = np.concatenate([orc.policy_forward_torch(sdt, xyz[i:i + 32], torch.from_numpy(qn[i:i + 32])).numpy()
                                 for i in range(0, xyz.shape[0], 32)])
    else:
        dq, _ = orc.policy_forward(sd, xyz, qn)
    q = np.clip(qn + dq, -1, 1).astype(np.float32)
    qu = orc.unnormalize(q, ft.JOINT_LIMITS_REAL)
    T = orc.franka_fk(qu)
    orc.transform_table(T, tp, tl, np.arange(2048, dtype=np.int32))
    centres = orc.transform_table(T, c, l)
    orc.collision_flags(centres[:, None], r, (prim["cuboid_centers"], prim["cuboid_dims"], prim["cuboid_quats"]),
                        (prim["cylinder_centers"], prim["cylinder_radii"], prim["cylinder_heights"],
                         prim["cylinder_quats"]))


def _cpu_worker(args):
    """One worker process of the CPU baseline: `threads` threads, environments [i0, i1) of the sample file."""
    path, i0, i1, threads, barrier = args
    from mpinets_amd import franka_tables as ft
    from oracle import oracle as orc

    data = np.load(path)
    sd = {k[3:]: data[k] for k in data.files if k.startswith("sd.")}
    prim = {k[5:]: data[k][i0:i1] for k in data.files if k.startswith("prim.")}
    xyz, qn = data["xyz"][i0:i1], data["qn"][i0:i1]
    tables = (ft.collision_sphere_table(False)[:3], ft.link_point_table())
    orc.set_threads(threads)
    cpu_step(orc, ft, sd, xyz[:2], qn[:2], {k: v[:2] for k, v in prim.items()}, tables, torch_threads=threads)  # (pools up)
    barrier.wait()
    t0 = time.time()
    cpu_step(orc, ft, sd, xyz, qn, prim, tables, torch_threads=threads)
    return t0, time.time()


def cpu_baseline(prob, model, n_env: int):
    """Oracle (CPU port) timed on `n_env` env-steps of the same workload on ALL host cores.  One process
    does not use 256 cores on this op mix (measured on the box: the torch fp32 restatement peaks at 16-32 threads, 22
    env-steps/s, and drops to 3 with 256), so the sample is cut over host_cores / 16 worker processes of 16 threads each
    (C parts OpenMP-parallel over environments, grouping / MLPs / heads as the oracle's torch restatement in fp32);
    `cores` = processes x threads.  The workers start together behind a barrier; the time is first start -> last end.
    `scalar_1t`: the same step on 8 environments in this process with the C parts on ONE thread and the float64 numpy
    MLPs (what rounds 1-3 reported)."""
    import multiprocessing as mp
    import tempfile

    from mpinets_amd import franka_tables as ft
    from oracle import oracle as orc

    orc.build()
    host = int(os.cpu_count() or 1)
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    tables = (ft.collision_sphere_table(False)[:3], ft.link_point_table())
    prim_keys = [k for k in prob if k.startswith(("cuboid_", "cylinder_"))]
    n1 = min(8, n_env)
    orc.set_threads(1)
    t0 = time.perf_counter()
    cpu_step(orc, ft, sd, prob["xyz"][:n1].cpu().numpy(), prob["q_norm"][:n1].cpu().numpy(),
             {k: prob[k][:n1].cpu().numpy() for k in prim_keys}, tables)
    dt1 = time.perf_counter() - t0
    threads = min(16, host)
    nw = max(1, min(host // threads, n_env // 8))
    cuts = [n_env * i // nw for i in range(nw + 1)]
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "sample.npz")
        np.savez(path, xyz=prob["xyz"][:n_env].cpu().numpy(), qn=prob["q_norm"][:n_env].cpu().numpy(),
                 **{"sd." + k: v for k, v in sd.items()}, **{"prim." + k: prob[k][:n_env].cpu().numpy() for k in prim_keys})
        ctx = mp.get_context("spawn")  # (fork is unsafe once the GPU runtime is up)
        with ctx.Manager() as mgr:
            barrier = mgr.Barrier(nw)
            with ctx.Pool(nw) as pool:
                spans = pool.map(_cpu_worker, [(path, cuts[i], cuts[i + 1], threads, barrier) for i in range(nw)])
    dt = max(e for _, e in spans) - min(s for s, _ in spans)
    return {"value": n_env / dt, "unit": "env-steps/s", "cores": nw * threads, "processes": nw, "threads_per_process": threads,
            "host_cores": host, "kind": "port",
            "scalar_1t": {"value": n1 / dt1, "unit": "env-steps/s", "cores": 1,
                          "sample": f"{n1} env-steps, C parts on one thread, float64-accumulate numpy MLPs, {dt1:.1f} s "
                                    "(the definition rounds 1-3 reported)"},
            "sample": f"{n_env} env-steps of the same step over {nw} processes x {threads} threads (oracle/: C FPS + ball query "
                      f"+ FK + SDF OpenMP-parallel over environments; grouping, MLPs and heads as the oracle's torch restatement "
                      f"in fp32), {dt:.1f} s from the first worker's start to the last one's end"}


HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E ~8 TB/s
FP32_VALU_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 vector peak (64 FLOP / clk / SIMD)
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak at 2.4 GHz
TRAFFIC_RECORD = "r06_traffic.json"  # written by tools/pmc_traffic.py from the PMC passes of this round
FAST_RECORD = "r06_fast_traffic.json"  # the same for the bf16x3 kernels (tools/pmc_fast.py)

# FLOPs per (collision sphere, unmasked primitive) pair, counted from csrc/sdf_device.h (one fma = 2): cuboid =
# projection 18 + 3 abs-sub + 3 max + 5 (norm) + sqrt + 3 (max3, min) + 2 (add, min-select) = 35; cylinder = 18 + 4 (rho)
# + 2 sub + 2 max + 3 + sqrt + 2 + 1 = 33; per sphere the FK-frame transform 18; per (env, waypoint) the 7-joint FK ~600
COL_CUB_FLOPS, COL_CYL_FLOPS, COL_SPHERE_FLOPS, COL_FK_FLOPS = 35, 33, 18, 600


def collision_flops(prob, rows, T: int, S: int = 56) -> float:
    """EXECUTED arithmetic of one swept-sphere collision call over environments `rows` x T waypoints: zero-volume
    (masked) primitives are skipped wave-uniformly by the kernel and are not counted."""
    live_c = int((prob["cuboid_dims"][rows].abs() > 1e-8).all(-1).sum().item())
    live_y = int(((prob["cylinder_radii"][rows].abs() > 1e-8) & (prob["cylinder_heights"][rows].abs() > 1e-8)).sum().item())
    n = prob["cuboid_dims"][rows].size(0)
    return float(T * S * (live_c * COL_CUB_FLOPS + live_y * COL_CYL_FLOPS) + n * T * (S * COL_SPHERE_FLOPS + COL_FK_FLOPS))


def kernel_source_hash(files=("sa_mlp.hip", "common.h")) -> str:
    """SHA-256 over kernel sources (default: the dominant kernel's, csrc/sa_mlp.hip + common.h): stamps a PMC record."""
    import hashlib

    h = hashlib.sha256()
    for f in files:
        h.update(open(os.path.join(ROOT, "motion-policy-networks_amd", "csrc", f), "rb").read())
    return h.hexdigest()


FAST_SOURCES = ("sa_mlp_bf16.hip", "dense_bf16.hip", "sa3_front_bf16.hip", "common.h")


def stage_table(prof, steps, B, sa1_ms, sa1_exec, sa2_ms, sa2_exec, col_flops):
    """achieved / peak per stage of the step: HBM stages in GB/s of algorithmic bytes, MFMA stages in TFLOP/s of
    executed FLOPs, FPS / ball query in G point-pair distance evaluations per second (VALU / latency bound: their
    HBM traffic, the 100 KB slab read once, is negligible)."""
    ms = lambda k: float(np.sum(prof[k])) / steps
    lin_ms = ms("mpx_linear") + ms("mpx_linear_ws") + ms("mpx_linear_rowmax")
    sa3_ms = ms("mpx_sa3_chain")  # the group-all module as one kernel (B >= 256); 0 when it ran layer by layer
    sa3_flops = 2.0 * B * 128 * (272 * 512 + 512 * 512 + 512 * 1024)  # (272 = 259 padded to whole slabs: executed work)
    lin_flops = 2.0 * B * (512 * 68 * 128 + 128 * 4 * 128            # SA2 first layer, per point / per query
                           + (0 if sa3_ms > 0 else 128 * (260 * 512 + 512 * 512 + 512 * 1024))  # group-all module
                           + 1024 * 4096 + 4096 * 2048 + 2048 * 2048    # fc head
                           + 8 * 32 + 32 * 64 + 64 * 128 + 128 * 128 + 128 * 64  # q encoder
                           + 2112 * 512 + 512 * 256 + 256 * 128 + 128 * 7)       # decoder
    pairs = B * (511 * 6272 + 127 * 512)
    bq_pairs = B * (512 * 6272 + 128 * 512)
    hbm = lambda by, t: {"bound": "hbm", "ms": t, "achieved": by / (t * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": by / (t * 1e-3) / 1e9 / HBM_PEAK_GBS}
    mf = lambda fl, t: {"bound": "mfma", "ms": t, "achieved": fl / (t * 1e-3) / 1e12, "peak": FP32_MFMA_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": fl / (t * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS}
    return {
        "fk_robot_cloud": dict(hbm(B * 2048 * 12.0, ms("mpx_franka_cloud")), note="24,576 B written per env"),
        "sphere_sdf_collision": {"bound": "valu", "ms": ms("mpx_franka_collision"),
                                 "achieved": col_flops / (ms("mpx_franka_collision") * 1e-3) / 1e12, "peak": FP32_VALU_PEAK_TFLOPS,
                                 "unit": "TFLOP/s", "frac": col_flops / (ms("mpx_franka_collision") * 1e-3) / 1e12 / FP32_VALU_PEAK_TFLOPS,
                                 "note": "executed FLOPs (unmasked primitives only) vs the fp32 VALU peak; one waypoint per "
                                         "environment here (56 of 64 lanes, one wave per environment): the T = 50 shape is "
                                         "extra_configs.c4_collision_validation"},
        "fps": {"bound": "valu/latency", "ms": ms("mpx_fps"), "achieved": pairs / (ms("mpx_fps") * 1e-3) / 1e9,
                "unit": "G point-distance updates/s", "note": "511 + 127 dependent picks per env"},
        "ball_query": {"bound": "valu", "ms": ms("mpx_ball_query"),
                       "achieved": bq_pairs / (ms("mpx_ball_query") * 1e-3) / 1e9, "unit": "G point-pair tests/s"},
        "sa1_grouped_mlp": mf(sa1_exec, sa1_ms), "sa2_grouped_mlp": mf(sa2_exec, sa2_ms),
        "sa3_group_all_chain": (dict(mf(sa3_flops, sa3_ms), note="mpx_sa3_chain: 3 layers + max-pool of the group-all module in "
                                     "one kernel, activations in LDS; nothing between the input rows and the pooled row in HBM")
                                if sa3_ms > 0 else None),
        "dense_layers": dict(mf(lin_flops, lin_ms), note="SA2 layer 1 (factored) + heads" +
                             ("" if sa3_ms > 0 else " + group-all module (layer by layer)")),
        "scene_cloud_rerender": dict(hbm(B * 4096 * 12.0, ms("mpx_scene_cloud")),
                                     note="49,152 B written per env; selection (Philox keys + radix select + counting sort in LDS, one workgroup per env) is the long pole, not HBM"),
        "groupnorm_leaky": {"bound": "hbm", "ms": ms("mpx_groupnorm_leaky")},
        "joint_step": {"bound": "latency", "ms": ms("mpx_joint_step")},
    }


def fast_roofline(B, live_ms, live_flops):
    """`roofline` of the bf16x3 mode's dominant kernel (sa2_bf16x3_persistent_kernel): achieved = the bf16 MFMA FLOPs of
    the live run / its live time; clock, pipe-busy fraction and HBM traffic from the committed PMC record -- dropped
    (null) when the kernel sources have changed since the passes were taken."""
    out = {"kernel": "sa2_bf16x3_persistent_kernel", "bound": "mfma", "achieved": live_flops / (live_ms * 1e-3) / 1e12,
           "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": live_flops / (live_ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS,
           "ms_per_launch": live_ms, "flops_per_launch": live_flops, "clock_ghz": None, "mfma_busy_frac": None,
           "frac_of_clock_adjusted_peak": None, "traffic": None}
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", FAST_RECORD)))
        if rec["envs_per_gpu"] == B and rec["kernel_source_sha256"] == kernel_source_hash(FAST_SOURCES):
            k = rec["kernels"]["sa2_bf16x3_persistent_kernel"]
            peak_at_clock = BF16_MFMA_PEAK_TFLOPS * k["clock_ghz"] / 2.4
            out.update(clock_ghz=k["clock_ghz"], mfma_busy_frac=k["mfma_busy_frac"], traffic=k["hbm_bytes_per_launch"],
                       frac_of_clock_adjusted_peak=out["achieved"] / peak_at_clock,
                       counters=f"profiles/{FAST_RECORD} (rocprofv3 --pmc passes at {B} envs; clock = GRBM_GUI_ACTIVE / 8 / "
                                "kernel time under the profiler; busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles))")
        else:
            out["counters"] = f"profiles/{FAST_RECORD} is stale (kernel source or batch size changed since the PMC passes)"
    except Exception as e:
        out["counters"] = f"no PMC record: {e}"
    return out


# multiply-adds per (query, neighbour) row of SA2 inside the fused kernel: layers 2 and 3 (128x128 + 128x256);
# layer 1 (67x128) runs once per point / per query as plain GEMMs (mpx_sa_mlp_factored)
SA2_ROW_MACS = 128 * 128 + 128 * 256
# multiply-adds per environment of the fc head, the joint encoder and the decoder (model.py:41-66, 385-393)
HEAD_MACS = (1024 * 4096 + 4096 * 2048 + 2048 * 2048 + 7 * 32 + 32 * 64 + 64 * 128 + 128 * 128 + 128 * 64
             + 2112 * 512 + 512 * 256 + 256 * 128 + 128 * 7)


def spawn_ranks_if_needed(args) -> bool:
    """`python bench.py --gpus N` (N > 1) outside a torchrun job: start the N ranks ourselves and exit with their
    status.  Returns True when the ranks of THIS job share GPUs (development mode, MPX_SHARE_GPU=1)."""
    shared = os.environ.get("MPX_SHARED_DEVICES") == "1"
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return shared
    import socket
    import subprocess

    ndev = torch.cuda.device_count()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this driver (RCCL needs it)
    if ndev < args.gpus:
        if os.environ.get("MPX_SHARE_GPU") != "1" or ndev == 0:
            sys.exit(f"bench.py: --gpus {args.gpus} but only {ndev} GPU(s) visible -- refusing to report a "
                     f"{args.gpus}-GPU number from fewer devices (development: MPX_SHARE_GPU=1 shares them over gloo)")
        env["MPX_SHARED_DEVICES"] = "1"
        env["MPX_DIST_BACKEND"] = "gloo"  # RCCL refuses two ranks on one device
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--envs", type=int, default=8192, help="environments per GPU")
    ap.add_argument("--fast-steps", type=int, default=3, help="steps of the secondary bf16x3 measurement (0 = skip)")
    ap.add_argument("--extra", type=int, default=1, help="also time BASELINE configs 2 and 4 (collision validation only)")
    ap.add_argument("--static-steps", type=int, default=2, help="steps of the extra without scene re-render on tabletop-only scenes (BASELINE config 3 shape at this batch size); 0 = skip")
    ap.add_argument("--all-slots-steps", type=int, default=2, help="steps of the worst-case extra: padding elision off, all 128 slots per neighbourhood (0 = skip)")
    ap.add_argument("--whole-batch-steps", type=int, default=0, help="opt-in extra: steps of the whole 65 536-environment configs[4] batch on this one GPU (0 = skip; 17 GB: the encoder works in slabs of 8192 environments)")
    ap.add_argument("--cpu-envs", type=int, default=512, help="env-steps in the CPU baseline sample (0 = skip)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak (default): --envs environments per GPU; strong: --global-envs environments split evenly over the GPUs")
    ap.add_argument("--global-envs", type=int, default=8192, help="total environments of a --scaling strong run")
    ap.add_argument("--train-steps", type=int, default=10, help="timed steps of the row-N1 training-step extra (0 = skip); the figure is the MEDIAN step, after 3 untimed ones")
    ap.add_argument("--scene-pool", type=int, default=1024, help="distinct host-generated primitive sets tiled over the batch (clouds are drawn per env on the device)")
    args = ap.parse_args()
    shared_devices = spawn_ranks_if_needed(args)

    from mpinets_amd import _lib, shard
    from mpinets_amd.model import MotionPolicyNetwork
    from mpinets_amd.rollout import RolloutEngine
    from mpinets_amd.scenes import make_problem_batch

    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    ndev = torch.cuda.device_count()
    local = int(os.environ.get("LOCAL_RANK", 0))
    # (shared_devices: development only -- several ranks on one GPU to exercise the N > 1 path)
    dev = torch.device("cuda", local % ndev if shared_devices else local)
    rank, ws, local = shard.init(device=dev)
    if ws != args.gpus:
        sys.exit(f"bench.py: WORLD_SIZE={ws} but --gpus {args.gpus} (launch with --nproc-per-node {args.gpus}, or run "
                 f"`python bench.py --gpus {args.gpus}` and let it start its own ranks)")
    n_gpus = ws
    _lib.load()

    # weak scaling (default): every rank owns --envs environments; strong: ONE fixed batch of --global-envs environments
    # cut into contiguous near-equal ranges (shard.split_even)
    strong = args.scaling == "strong"
    envs = shard.split_even(args.global_envs, n_gpus)[rank] if strong else shard.env_range(rank, n_gpus, args.envs)
    B = len(envs)
    global_envs = args.global_envs if strong else args.envs * n_gpus
    assert B > 0, f"--scaling strong: {args.global_envs} environments cannot feed {n_gpus} ranks"
    torch.manual_seed(0)  # identical random-init weights on every rank (replicated, like a checkpoint)
    model = MotionPolicyNetwork().to(dev).eval()
    # ONE global batch of n_gpus * B problems that depends on the seed only; this rank owns rows `envs` of it and every
    # random draw (scene clouds at set-up and at every re-render) is keyed by the GLOBAL environment id, so the
    # gathered result of an N-rank run equals a single-rank run over the same environments bit for bit
    # (tests/test_gpu_shard.py)
    prob = make_problem_batch(B, seed=1000, device=dev, kinds=("tabletop", "cubby", "dresser"), M1=40, M2=16,
                              scene_pool=args.scene_pool, device_clouds=True, env_offset=envs.start,
                              total_envs=global_envs)
    # the reference's loop (model.py:170-181, run_inference.py:188-189): the robot cloud's column subset is redrawn at
    # EVERY step, one draw for the batch -- on the device here (mpx_draw_subset), the same on every rank
    eng = RolloutEngine(model, prob, rerender_scene=True, scene_seed=17, resample_subset=True, subset_seed=23)

    for _ in range(args.warmup):
        eng.step()
    torch.cuda.synchronize()
    shard.barrier()
    _lib.profile_start("mpx_sa_mlp", "mpx_sa_mlp_factored", "mpx_fps", "mpx_ball_query", "mpx_linear", "mpx_linear_ws", "mpx_linear_rowmax",
                        "mpx_franka_cloud", "mpx_franka_collision", "mpx_joint_step", "mpx_groupnorm_leaky", "mpx_scene_cloud",
                        "mpx_sa3_chain", "mpx_draw_subset")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.step()
    torch.cuda.synchronize()
    shard.barrier()
    elapsed = time.perf_counter() - t0
    prof = _lib.profile_stop()
    elapsed_local = elapsed
    elapsed = shard.max_over_ranks(elapsed, dev)
    # self-check record of an N-rank run: which communicator, which device every rank ran on, every rank's own time
    props = torch.cuda.get_device_properties(dev)
    rank_records = shard.gather_objects({
        "rank": rank, "local_rank": local, "device": str(dev), "device_name": props.name,
        "gcn_arch": getattr(props, "gcnArchName", ""), "pci_bus_id": getattr(props, "pci_bus_id", None),
        "env_ids": [envs.start, envs.stop], "ms_per_step": elapsed_local / args.steps * 1e3})
    dist_backend = shard.backend_name()
    cnt1, cnt2 = (c.clone() for c in model.point_cloud_encoder.last_counts)  # ball-query hit counts of the last timed step

    # ---- secondary measurement: the opt-in split-bf16 mode of the two grouped MLPs (same step, same
    # envs, continuing the rollout).  The headline above stays the exact-fp32 path.
    fast = None
    if args.fast_steps > 0:
        model.set_precision("bf16x3")
        eng.step()  # packs the bf16 operand blocks + warm-up
        torch.cuda.synchronize()
        shard.barrier()
        dense_calls = ("mpx_linear_bf16x3", "mpx_linear_rowmax_bf16x3", "mpx_linear_bf16x3_to_pairs", "mpx_linear_bf16x3_pairs",
                       "mpx_linear_rowmax_bf16x3_pairs", "mpx_sa3_front_bf16x3", "mpx_linear", "mpx_linear_ws")
        _lib.profile_start("mpx_sa_mlp_bf16x3", "mpx_sa_mlp_bf16x3_factored", *dense_calls)
        tf0 = time.perf_counter()
        for _ in range(args.fast_steps):
            eng.step()
        torch.cuda.synchronize()
        shard.barrier()
        fel = shard.max_over_ranks(time.perf_counter() - tf0, dev)
        fall = _lib.profile_stop()
        model.set_precision("fp32")
        fdense = float(sum(np.sum(fall[k]) for k in dense_calls)) / args.fast_steps
        fast = (fel, float(np.mean(fall["mpx_sa_mlp_bf16x3"])), float(np.mean(fall["mpx_sa_mlp_bf16x3_factored"])), fdense)

    # final host gather (the only cross-rank data movement): joint angles + collision flags
    q_all = shard.gather_to_rank0(eng.q)
    f_all = shard.gather_to_rank0(eng.flags)

    # ---- extra: BASELINE configs 3 and 1 ("8192 envs x 50-waypoint expert-trajectory collision validation, sharded" and
    # "1024 envs FK + swept-sphere SDF") -- EVERY rank on its own shard, inside barriers: config 3's 8192 trajectories are
    # split evenly over the ranks (its definition is a fixed global batch), config 1 runs 1024 environments per rank
    extra = {}
    collision = None
    if args.extra:
        from mpinets_amd.geometry import TorchCuboids, TorchCylinders
        from mpinets_amd.scenes import linear_trajectories, random_configurations

        def timed(fn, n=20):
            fn()
            torch.cuda.synchronize()
            shard.barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n

        C4_ENVS = 8192
        mine4 = shard.split_even(C4_ENVS, n_gpus)[rank]
        n4 = min(len(mine4), B)  # (a development run with a small --envs validates what it has scenes for)
        n2 = min(1024, B)
        traj = torch.from_numpy(linear_trajectories(C4_ENVS, 50, 5)[mine4.start:mine4.start + n4]).to(dev)
        q1k = torch.from_numpy(random_configurations(1024 * n_gpus, 6)[rank * 1024:rank * 1024 + n2]).to(dev)
        sub = lambda t, n: t[:n].contiguous()
        prims = lambda n: (TorchCuboids(sub(prob["cuboid_centers"], n), sub(prob["cuboid_dims"], n), sub(prob["cuboid_quats"], n)),
                           TorchCylinders(sub(prob["cylinder_centers"], n), sub(prob["cylinder_radii"], n),
                                          sub(prob["cylinder_heights"], n), sub(prob["cylinder_quats"], n)))
        cub4, cyl4 = prims(n4)
        cub2, cyl2 = prims(n2)
        c4_ms = timed(lambda: eng.collision.check(traj, cub4, cyl4))
        c2_ms = timed(lambda: eng.collision.check(q1k, cub2, cyl2, return_sdf=True))
        flags4 = shard.gather_to_rank0(eng.collision.check(traj, cub4, cyl4).to(torch.int32))  # (the final host gather of config 3)
        col_recs = shard.gather_objects({"rank": rank, "c4_envs": n4, "c4_ms": c4_ms, "c4_gflop": collision_flops(prob, slice(0, n4), 50) / 1e9,
                                         "c2_envs": n2, "c2_ms": c2_ms, "c2_gflop": collision_flops(prob, slice(0, n2), 1) / 1e9})
        collision = {"records": col_recs, "c4_rate": None if flags4 is None else float((flags4 != 0).float().mean())}
        del traj, q1k, cub4, cyl4, cub2, cyl2

    # ---- extra: the same closed-loop step WITHOUT the scene re-render, on tabletop-only scenes (16 cuboids + 16
    # cylinders): the reference's rollout re-samples only the robot points (model.py:180-181); all ranks, weak scaling
    static = None
    if args.extra and args.static_steps > 0:
        prob_s = make_problem_batch(B, seed=5000, device=dev, kinds=("tabletop",), M1=16, M2=16,
                                    scene_pool=args.scene_pool, device_clouds=True, env_offset=envs.start,
                                    total_envs=global_envs)
        eng_s = RolloutEngine(model, prob_s)
        eng_s.step()
        torch.cuda.synchronize()
        shard.barrier()
        names_s = ("mpx_sa_mlp", "mpx_sa_mlp_factored", "mpx_fps", "mpx_ball_query", "mpx_linear", "mpx_linear_ws", "mpx_linear_rowmax",
                   "mpx_sa3_chain")
        _lib.profile_start(*names_s)
        ts0 = time.perf_counter()
        for _ in range(args.static_steps):
            eng_s.step()
        torch.cuda.synchronize()
        shard.barrier()
        el_s = shard.max_over_ranks(time.perf_counter() - ts0, dev)
        prof_s = _lib.profile_stop()
        static = {"envs_per_gpu": B, "steps": args.static_steps, "ms_per_step": el_s / args.static_steps * 1e3,
                  "env_steps_per_s": global_envs * args.static_steps / el_s, "dtype": "f32",
                  "collision_rate": float((eng_s.flags != 0).float().mean().item()),
                  "kernels_ms_per_step": {k[4:]: float(np.sum(v)) / args.static_steps for k, v in prof_s.items()},
                  "mean_distinct_neighbours": [float(c.float().mean().item()) for c in model.point_cloud_encoder.last_counts],
                  "what": "tabletop scenes (16 cuboids + 16 cylinders, zero-padded), static scene cloud; every step: policy "
                          "forward + joint update + FK cloud refresh + collision check"}
        del eng_s, prob_s
        extra["tabletop_static_scene"] = static

    # ---- extra: the worst case of the headline -- the SAME workload with the padding elision switched off: every
    # neighbourhood walks its nominal 128 slots like the reference does (model.set_elide_padding(False))
    all_slots = None
    if args.extra and args.all_slots_steps > 0:
        model.set_elide_padding(False)
        eng.step()
        torch.cuda.synchronize()
        shard.barrier()
        ta0 = time.perf_counter()
        for _ in range(args.all_slots_steps):
            eng.step()
        torch.cuda.synchronize()
        shard.barrier()
        el_a = shard.max_over_ranks(time.perf_counter() - ta0, dev)
        model.set_elide_padding(True)
        all_slots = {"steps": args.all_slots_steps, "ms_per_step": el_a / args.all_slots_steps * 1e3,
                     "env_steps_per_s": global_envs * args.all_slots_steps / el_a, "dtype": "f32",
                     "what": "the headline workload with padding elision OFF: the grouped MLPs evaluate all 128 slots of every "
                             "neighbourhood (what the reference computes; same result bit for bit) -- the density-independent "
                             "floor of the headline number"}

    # ---- extra: row N1, one optimisation step (forward + losses + backward + gradient all-reduce over RCCL + clip + Adam;
    # model.py:185-240, run_training.py:71-77) at the reference's batch size per GPU (jobconfig.yaml: 10) and at 256 --
    # every rank, data-parallel: the ONE collective of the code base
    training = None
    if args.extra and args.train_steps > 0:
        from mpinets_amd.model import TrainingMotionPolicyNetwork
        from mpinets_amd.training import train_step

        scene_keys = ("cuboid_centers", "cuboid_dims", "cuboid_quats", "cylinder_centers", "cylinder_radii",
                      "cylinder_heights", "cylinder_quats")
        torch.manual_seed(0)
        tm = TrainingMotionPolicyNetwork(2048, 1.0, 5.0).to(dev)  # (loss weights of jobconfig.yaml)
        opt = tm.configure_optimizers()
        training = {}
        for tb, prec in ((10, "fp32"), (256, "fp32"), (256, "bf16x3")):
            nb = min(tb, B)
            tm.set_training_precision(prec)
            g = torch.Generator(device="cpu").manual_seed(100 + rank)
            sup = torch.clamp(prob["q_norm"][:nb] + 0.05 * torch.randn(nb, 7, generator=g).to(dev), -1, 1)
            batch = {"xyz": prob["xyz"][:nb].clone(), "configuration": prob["q_norm"][:nb].clone(), "supervision": sup,
                     **{k: prob[k][:nb] for k in scene_keys}}
            # Three untimed steps size the caching allocator's pools for this batch (the first steps at a new batch size
            # pay hipMalloc for every activation: the round-5 driver record took 42 ms / step over 3 timed steps where the
            # settled step is 22 ms); every timed step is then bracketed by its own pair of HIP events on the step's
            # stream, so one stalled step shows as `ms_max` instead of moving the figure.  `ms_per_step` = MEDIAN step,
            # max over ranks; `ms_wall_mean` = wall clock / steps around the same loop (host gaps included).
            for _ in range(3):
                train_step(tm, opt, batch)
            torch.cuda.synchronize()
            shard.barrier()
            n_t = args.train_steps
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_t)]
            tt0 = time.perf_counter()
            for e0, e1 in evs:
                e0.record()
                loss = train_step(tm, opt, batch)
                e1.record()
            torch.cuda.synchronize()
            shard.barrier()
            wall_t = shard.max_over_ranks(time.perf_counter() - tt0, dev)
            per_step = sorted(e0.elapsed_time(e1) for e0, e1 in evs)  # ms
            el_t = shard.max_over_ranks(per_step[len(per_step) // 2] * 1e-3, dev) * n_t  # (median step) x steps
            ms_max = shard.max_over_ranks(per_step[-1] * 1e-3, dev) * 1e3
            c1t, c2t = tm.point_cloud_encoder.last_counts
            rows1, rows2 = int(c1t.clamp(min=1).sum()), int(c2t.clamp(min=1).sum())
            # matrix work: forward over the hit rows only (the differentiable path packs them); a dense backward is 2x that
            # (`nominal`); the backward of the three pooled layers walks the pool's non-zero gradients instead (mpx_pool_wgrad /
            # mpx_pool_dgrad: Q x C x K fused multiply-adds each on the vector unit), so `executed` drops their two GEMMs
            fwd = 2.0 * (rows1 * 8448 + rows2 * 57728 + nb * (128 * 919040 + HEAD_MACS))
            pooled_dense = 2.0 * 2 * (rows1 * 4096 + rows2 * 32768 + nb * 128 * 524288)
            pooled_sparse = 2.0 * 2 * (c1t.numel() * 64 * 64 + c2t.numel() * 256 * 128 + nb * 1024 * 512)
            executed = 3 * fwd - pooled_dense + pooled_sparse
            training[f"batch_{tb}" + ("" if prec == "fp32" else "_" + prec)] = {
                "samples_per_gpu": nb, "steps": n_t, "ms_per_step": el_t / n_t * 1e3, "ms_min": per_step[0],
                "ms_max": ms_max, "ms_wall_mean": wall_t / n_t * 1e3,
                "samples_per_s": nb * n_gpus * n_t / el_t, "loss": float(loss.item()), "dtype": prec,
                "nominal_tflops": 3 * fwd / (el_t / n_t) / 1e12, "executed_tflops": executed / (el_t / n_t) / 1e12,
                "frac_of_fp32_mfma_peak": executed / (el_t / n_t) / 1e12 / FP32_MFMA_PEAK_TFLOPS}
        tm.set_training_precision("fp32")
        training["what"] = ("TrainingMotionPolicyNetwork.training_step + backward + bucketed gradient all-reduce + clip(1.0) + Adam "
                            "(mpinets_amd.training.train_step); batch_10 = jobconfig.yaml's batch size per GPU; batch_256_bf16x3 = the same step "
                            "with set_training_precision('bf16x3') (the grouped MLPs' GEMMs in split bf16, fp32 accumulate and master "
                            "weights: the engine's form of the reference's precision=16); nominal FLOPs = 3 x the forward's matrix "
                            "work over the hit rows; executed = that minus the dense backward of the three pooled layers, which "
                            "runs over the max-pool's non-zero gradients; the fraction is executed work against the fp32 MFMA peak")
        training["allreduce_ranks"] = n_gpus if dist_backend is not None else 1
        del tm, opt, batch
        model.eval()
        extra["n1_training_step"] = training

    # ---- every cross-rank measurement is done: release the other ranks.  What follows runs on rank 0 alone (single-GPU
    # configurations and the host-side CPU baseline) while nobody waits at a barrier.
    shard.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    if rank != 0:
        return

    # ---- rank 0: BASELINE configs 0 and 2 -- the same closed-loop step (static scene) for a single problem and for
    # 256 problems x 50 steps
    if args.extra:
        small = {}
        for nb, nsteps, prec in ((1, 50, "fp32"), (256, 50, "fp32"), (256, 50, "bf16x3")):
            ps = make_problem_batch(nb, seed=7000 + nb, device=dev, kinds=("tabletop",), M1=16, M2=16, scene_pool=64,
                                    device_clouds=True)
            model.set_precision(prec)
            es = RolloutEngine(model, ps, resample_subset=True, subset_seed=23)
            es.step()
            torch.cuda.synchronize()
            t_s = time.perf_counter()
            for _ in range(nsteps):
                es.step()
            torch.cuda.synchronize()
            small[nb if prec == "fp32" else "256x3"] = (time.perf_counter() - t_s) * 1e3
            del es, ps
        model.set_precision("fp32")
        extra["c1_single_problem"] = {
            "envs": 1, "steps": 50, "ms_per_step": small[1] / 50, "rollout_ms": small[1],
            "what": "one tabletop problem, 50 closed-loop steps (the reference's deployed use; it assumes 80 ms per step, "
                    "run_inference.py:297)"}
        extra["c3_rollout_256"] = {
            "envs": 256, "steps": 50, "ms_per_step": small[256] / 50, "rollout_ms": small[256],
            "env_steps_per_s": 256 * 50 / small[256] * 1e3, "dtype": "f32",
            "bf16x3_rollout_ms": small["256x3"], "bf16x3_env_steps_per_s": 256 * 50 / small["256x3"] * 1e3,
            "what": "256 tabletop problems, 50-step rollout (policy forward + joint update + FK cloud refresh + collision "
                    "check per step)",
            "precision_note": "BASELINE configs[2] names bf16 for the policy forward; plain bf16 products miss the north "
                              "star's 1e-5 bar on the policy deltas (2.6e-4 simulated), so the bf16 matrix cores are offered "
                              "only as 'bf16x3' (3 split products per fp32 product, fp32 accumulate; 2.6e-7 measured) -- the "
                              "bf16x3_* fields are that mode, the headline fields exact fp32"}
    if collision is not None:
        recs = collision["records"]
        c4_envs, c4_ms = sum(r["c4_envs"] for r in recs), max(r["c4_ms"] for r in recs)
        c2_envs, c2_ms = sum(r["c2_envs"] for r in recs), max(r["c2_ms"] for r in recs)
        c4_flop, c2_flop = sum(r["c4_gflop"] for r in recs) * 1e9, sum(r["c2_gflop"] for r in recs) * 1e9
        c2_cpu = c4_cpu = None
        cpu_threads = 1
        if args.cpu_envs > 0:  # the same work on the host: oracle port on all cores (OpenMP over environments)
            from mpinets_amd import franka_tables as ft
            from mpinets_amd.scenes import linear_trajectories, random_configurations
            from oracle import oracle as orc

            orc.build()
            cpu_threads = orc.set_threads(int(os.cpu_count() or 1))
            c_, r_, l_, _ = ft.collision_sphere_table(False)
            host_prims = lambda n: {k: prob[k][:n].cpu().numpy() for k in prob if k.startswith(("cuboid_", "cylinder_"))}

            def host_check(qh, T, ph):
                n = ph["cuboid_centers"].shape[0]
                ctr = orc.transform_table(orc.franka_fk(qh.reshape(-1, 7)), c_, l_).reshape(n, T, -1, 3)
                orc.collision_flags(ctr, r_, (ph["cuboid_centers"], ph["cuboid_dims"], ph["cuboid_quats"]),
                                    (ph["cylinder_centers"], ph["cylinder_radii"], ph["cylinder_heights"], ph["cylinder_quats"]))

            n2h = min(1024, B)
            th = time.perf_counter()
            host_check(random_configurations(1024, 6)[:n2h], 1, host_prims(n2h))
            c2_cpu = n2h / (time.perf_counter() - th)
            n4h = min(512, B)
            th = time.perf_counter()
            host_check(linear_trajectories(8192, 50, 5)[:n4h], 50, host_prims(n4h))
            c4_cpu = n4h * 50 / (time.perf_counter() - th)
        extra["c4_collision_validation"] = {
            "envs": c4_envs, "waypoints": 50, "sharding": f"{c4_envs} trajectories split evenly over {n_gpus} rank(s), no collective, one final gather of the flags",
            "ms": c4_ms, "ms_per_rank": [r["c4_ms"] for r in recs], "envs_per_rank": [r["c4_envs"] for r in recs],
            "env_waypoints_per_s": c4_envs * 50 / c4_ms * 1e3, "collision_rate": collision["c4_rate"],
            "cpu_port_env_waypoints_per_s": c4_cpu, "cpu_cores": cpu_threads,
            "roofline": {"bound": "valu", "achieved": c4_flop / (c4_ms * 1e-3) / 1e12, "peak": FP32_VALU_PEAK_TFLOPS * n_gpus,
                         "unit": "TFLOP/s", "frac": c4_flop / (c4_ms * 1e-3) / 1e12 / (FP32_VALU_PEAK_TFLOPS * n_gpus),
                         "executed_gflop": c4_flop / 1e9,
                         "note": "executed FLOPs: unmasked primitives only (35 per sphere-cuboid, 33 per sphere-cylinder pair, "
                                 "csrc/sdf_device.h) vs the fp32 VALU peak of all ranks; counters: profiles/r06_collision_pmc_pass*.csv"},
            "what": "FK + 56-sphere SDF vs 40 cuboids + 16 cylinders (zero-padded), has_collision[B] (model.py:293-314)"}
        extra["c2_fk_sdf_1024"] = {
            "envs": c2_envs, "envs_per_rank": [r["c2_envs"] for r in recs], "ms": c2_ms, "ms_per_rank": [r["c2_ms"] for r in recs],
            "env_steps_per_s": c2_envs / c2_ms * 1e3, "cpu_port_env_steps_per_s": c2_cpu, "cpu_cores": cpu_threads,
            "roofline": {"bound": "valu (launch / latency bound at this size)", "achieved": c2_flop / (c2_ms * 1e-3) / 1e12,
                         "peak": FP32_VALU_PEAK_TFLOPS * n_gpus, "unit": "TFLOP/s",
                         "frac": c2_flop / (c2_ms * 1e-3) / 1e12 / (FP32_VALU_PEAK_TFLOPS * n_gpus),
                         "note": "1024 environments per rank: a launch-latency-sized call"},
            "what": "FK + sphere SDF, flags + min-sdf [1024,56] written; 1024 environments on every rank"}

    # ---- extra (opt-in, rank 0): the WHOLE of BASELINE configs[4] -- 65 536 environments -- resident on ONE GPU (the
    # encoder works in slabs of 8192 environments that reuse one workspace: 17 GB of the 288 GB)
    whole = None
    if args.whole_batch_steps > 0 and not shared_devices:
        del eng
        torch.cuda.empty_cache()
        BW = 65536
        prob_w = make_problem_batch(BW, seed=1000, device=dev, kinds=("tabletop", "cubby", "dresser"), M1=40, M2=16,
                                    scene_pool=args.scene_pool, device_clouds=True)
        eng_w = RolloutEngine(model, prob_w, rerender_scene=True, scene_seed=17)
        eng_w.step()
        torch.cuda.synchronize()
        tw0 = time.perf_counter()
        for _ in range(args.whole_batch_steps):
            eng_w.step()
        torch.cuda.synchronize()
        el_w = time.perf_counter() - tw0
        whole = {"envs": BW, "steps": args.whole_batch_steps, "ms_per_step": el_w / args.whole_batch_steps * 1e3,
                 "env_steps_per_s": BW * args.whole_batch_steps / el_w, "dtype": "f32",
                 "peak_memory_gb": torch.cuda.max_memory_allocated(dev) / 1e9,
                 "collision_rate": float((eng_w.flags != 0).float().mean().item()),
                 "what": "BASELINE configs[4] whole (65 536 mixed environments, re-render + policy step) on ONE MI355X"}
        del eng_w, prob_w
        torch.cuda.empty_cache()

    # ---- rank 0: the JSON line
    # SA1 = mpx_sa_mlp; SA2 = mpx_sa_mlp_factored (first layer evaluated per point / per query by two
    # mpx_linear calls, which are timed under linear_all)
    sa1_ms, sa2_ms = float(np.mean(prof["mpx_sa_mlp"])), float(np.mean(prof["mpx_sa_mlp_factored"]))
    # Executed work: only the DISTINCT neighbours of a ball-query neighbourhood are evaluated (its padding
    # repeats the first neighbour; max-pooling is idempotent -> bit-identical output).  The roofline uses
    # the FLOPs of the 32-row MFMA tiles actually issued (counts of the last timed step), not the nominal
    # 128 slots per neighbourhood.

    def tiles(c, q, gr):  # fp32 kernel: q consecutive queries per wave, rows packed at gr-row granularity (csrc/sa_mlp.hip: GR)
        rows = (c.clamp(1, 128) + gr - 1) // gr * gr
        return int(((rows.reshape(-1, q).sum(1) + 31) // 32).sum().item())

    # queries per unit as the launchers choose them (csrc/sa_mlp.hip: launch_sa / mpx_sa_mlp_factored)
    q1 = 32 if cnt1.numel() >= 1024 * 32 else 4
    q2 = 8 if cnt2.numel() >= 1024 * 8 else (2 if cnt2.numel() >= 1024 else 1)
    t1, t2 = tiles(cnt1, q1, 2), tiles(cnt2, q2, 4)
    sa1_exec, sa2_exec = t1 * 32 * 8448 * 2, t2 * 32 * SA2_ROW_MACS * 2
    achieved = sa2_exec / (sa2_ms * 1e-3) / 1e12
    total_envsteps = global_envs * args.steps
    # HBM traffic of the dominant kernel comes from committed PMC passes (it cannot be read live).  The record
    # carries the SHA-256 of the kernel source it was measured on: a changed kernel file -> traffic null
    traffic, traffic_note, traffic_step = None, None, None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", TRAFFIC_RECORD)))
        if tj["envs_per_gpu"] == B and tj["kernel_source_sha256"] == kernel_source_hash():
            traffic, traffic_note = tj["hbm_bytes_per_launch"], tj["source"] + "; " + tj["correction"]
            traffic_step = tj.get("whole_step")  # every kernel of the step: corrected HBM bytes vs SURVEY 8(d)'s algorithmic bytes
        else:
            traffic_note = f"profiles/{TRAFFIC_RECORD} is stale (kernel source or batch size changed since the PMC passes)"
    except Exception as e:
        traffic_note = f"no PMC record: {e}"
    out = {
        "metric": "env-steps/sec (FK+SDF+PointNet++)",
        "value": total_envsteps / elapsed,
        "unit": "env-steps/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": (f"BASELINE configs[4] per-GPU share: closed-loop step over {B} mixed tabletop / cubby / dresser envs per GPU"
                         if not strong else
                         f"BASELINE configs[4] step, STRONG scaling: one batch of {global_envs} mixed tabletop / cubby / dresser "
                         f"envs split evenly over {n_gpus} GPU(s)") +
                        ": scene cloud re-render (4096 pts from the primitives) + PointNet++ "
                        "forward (2048 robot + 4096 scene + 128 target pts) + joint update + FK robot-cloud refresh with the "
                        "column subset redrawn every step (the reference's loop, model.py:170-181) + 56-sphere SDF collision "
                        "check vs 40 cuboids + 16 cylinders (zero-padded)",
            "envs_per_gpu": B, "global_envs": global_envs, "points_per_env": int(prob["xyz"].size(1)),
            "parallelism": f"env-sharded x{n_gpus}, no collective on the step",
            "weights": "random-init (seed 0)",
            "scene_pool": f"{min(args.scene_pool, global_envs)} distinct host-generated primitive sets tiled over the "
                          f"{global_envs} global environments (env g uses set g mod pool); every environment draws its "
                          "own scene cloud on the device, keyed by its global id",
            "env_ids": [envs.start, envs.stop] if n_gpus == 1 else [r["env_ids"] for r in rank_records],
            **({"devices_shared": True, "physical_gpus": ndev,
                "note": "DEVELOPMENT RUN: the ranks share GPUs (MPX_SHARE_GPU=1, gloo) -- not a scaling number"}
               if shared_devices else {}),
        },
        "roofline": {
            "kernel": "sa_mlp_packed_kernel<64,128,128,256,8,true> (SA2 fused group + MLP layers 2-3 + maxpool; layer 1 factored out)",
            "bound": "mfma", "achieved": achieved, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved / FP32_MFMA_PEAK_TFLOPS,
            # the same launch priced with SURVEY.md 8(d)'s NOMINAL work (all 128 slots of every neighbourhood, layer 1
            # per (query, neighbour) row: 1.892 GFLOP per env): above 1 means the kernel does not do the nominal work --
            # it skips ball-query padding and evaluates layer 1 per point.  Scene-density dependent; `frac` is the
            # executed-work fraction, `all_slots` the density-independent floor of the whole step.
            "frac_nominal": SA2_FLOPS * B / (sa2_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
            "traffic": traffic, "traffic_source": traffic_note, "traffic_whole_step": traffic_step,
            "ms_per_launch": sa2_ms, "flops_per_launch": sa2_exec,
            "note": "achieved = FLOPs of the 32-row MFMA tiles actually issued / time; ball-query padding "
                    "(repeats of the first neighbour) is not re-evaluated (bit-identical result)",
            "tiles_walked": t2, "tiles_nominal": B * 128 * 4, "nominal_flops_per_launch": SA2_FLOPS * B,
            "nominal_equivalent_tflops": SA2_FLOPS * B / (sa2_ms * 1e-3) / 1e12,
        },
        "kernels_ms": {
            "sa2_mlp": sa2_ms, "sa1_mlp": sa1_ms,
            "sa1_tflops_executed": sa1_exec / (sa1_ms * 1e-3) / 1e12, "sa1_tiles_walked": t1,
            "sa1_tiles_nominal": B * 512 * 4,
            "fps": float(np.sum(prof["mpx_fps"])) / args.steps,
            "ball_query": float(np.sum(prof["mpx_ball_query"])) / args.steps,
            "linear_all": float(np.sum(prof["mpx_linear"]) + np.sum(prof["mpx_linear_ws"]) + np.sum(prof["mpx_linear_rowmax"])) / args.steps,
            "sa3_chain": float(np.sum(prof["mpx_sa3_chain"])) / args.steps,
        },
        # per-stage achieved / peak with the ALGORITHMIC work of SURVEY.md section 8(d) (per env-step, x B envs)
        "stages": stage_table(prof, args.steps, B, sa1_ms, sa1_exec, sa2_ms, sa2_exec,
                              collision_flops(prob, slice(None), 1)),
        "result_check": {"gathered_q": list(q_all.shape), "collision_rate": float((f_all != 0).float().mean())},
        # N-rank self-check: the communicator the barrier / MAX / gather ran on ("nccl" = RCCL; null = a plain single
        # process, no group), the number of ranks on it, and every rank's device + own ms_per_step (`ms_per_step`
        # above is their maximum)
        "dist_backend": dist_backend, "rccl_ranks": n_gpus if dist_backend == "nccl" else 0,
        "ranks": rank_records,
    }
    if extra:
        out["extra_configs"] = extra
    if all_slots is not None:
        out["all_slots"] = all_slots
    if whole is not None:
        out["whole_config4_one_gpu"] = whole
    if fast is not None:
        fel, f1_ms, f2_ms, fdense_ms = fast
        out["fast_mode"] = {
            "what": "same step with the grouped MLPs and the large dense layers on the bf16 matrix cores, each fp32 product "
                    "evaluated as hi*hi + hi*lo + lo*hi (split-bf16, fp32 accumulate); opt-in via "
                    "model.set_precision('bf16x3'); policy deltas stay within 1e-5 of the fp32 oracle "
                    "(tests/test_gpu_policy.py: 2.6e-7 measured)",
            "dtype": "bf16x3", "value": global_envs * args.fast_steps / fel, "unit": "env-steps/s",
            "steps": args.fast_steps, "ms_per_step": fel / args.fast_steps * 1e3,
            "sa1_ms": f1_ms, "sa2_ms": f2_ms, "dense_ms": fdense_ms,
            # the persistent bf16x3 kernel packs the rows of 8 consecutive queries per unit, like the fp32 kernel:
            # the same tile count; three bf16 MFMAs per fp32 product
            "sa2_executed_tflops": t2 * 32 * SA2_ROW_MACS * 2 / (f2_ms * 1e-3) / 1e12,
            "sa2_bf16_mfma_tflops": 3 * t2 * 32 * SA2_ROW_MACS * 2 / (f2_ms * 1e-3) / 1e12,
            "sa2_frac_of_bf16_peak_2500": 3 * t2 * 32 * SA2_ROW_MACS * 2 / (f2_ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS,
            "roofline": fast_roofline(B, f2_ms, 3 * t2 * 32 * SA2_ROW_MACS * 2),
        }
    if args.cpu_envs > 0:  # rank 0's host cores, for every N (the other ranks were released above)
        out["cpu_baseline"] = cpu_baseline(prob, model, args.cpu_envs)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
