"""Shard == single rank on the REAL engine (SURVEY.md section 8e, section 4 item iv).

Two ranks (gloo, both on the one GPU of the box) each run the headline step -- RolloutEngine(rerender_scene=True):
scene re-render + policy forward + joint update + FK cloud refresh + collision check -- over their half of one
global batch; the gathered joint angles, collision flags and slab rows must equal a single-rank run over the whole
batch BIT FOR BIT.  That holds because every random draw is keyed by the global environment id
(mpx_scene_cloud's env_offset) and no kernel couples environments.

Batch sizes: the dense layers pick their launch shape from the row count (split-K below 1025 rows and few output
tiles, csrc/dense.hip splitk_plan), which changes the fp32 summation order; E and 2E are chosen inside one regime
(24 | 48: both split the same way; 1040 | 2080: neither splits), like the 8192-per-rank bench shards.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("E,steps", [(24, 2), (1040, 2)])
def test_two_ranks_equal_one_rank_bit_for_bit(tmp_path, E, steps):
    sys.path.insert(0, HERE)
    from shard_worker import run_range

    out = str(tmp_path / "sharded.npz")
    env = dict(os.environ, MPX_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(HERE, "shard_worker.py"), str(E), str(steps), out]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    got = dict(np.load(out))
    ref = run_range(torch.device("cuda:0"), 0, 2 * E, 2 * E, steps)
    assert got["q"].shape == (2 * E, 7)
    for k, v in ref.items():
        np.testing.assert_array_equal(got[k], v.cpu().numpy(), err_msg=k)
    # the two halves are different problems (the comparison is not vacuous)
    assert not np.array_equal(got["q"][:E], got["q"][E:])


def test_scene_draw_is_keyed_by_global_env_id(oracle):
    """mpx_scene_cloud(env_offset=o) on rows [o, o+n) == rows [o, o+n) of the unsharded call == the oracle."""
    from mpinets_amd.scenes import make_scenes, sample_scene_clouds

    dev = torch.device("cuda:0")
    scn = make_scenes(12, 3, ("tabletop", "cubby", "dresser"), 40, 16)
    prims = {k: torch.from_numpy(v).to(dev) for k, v in scn.items()}
    full, a_full, l_full, _ = sample_scene_clouds(prims, 4096, 99, write_label=True, return_aux=True)
    for o, n in ((0, 5), (5, 7), (11, 1)):
        part = {k: v[o:o + n].contiguous() for k, v in prims.items()}
        got, a, lab, _ = sample_scene_clouds(part, 4096, 99, write_label=True, return_aux=True, env_offset=o)
        assert torch.equal(got, full[o:o + n]) and torch.equal(a, a_full[o:o + n]) and torch.equal(lab, l_full[o:o + n])
        pts, assign, labels, _ = oracle.scene_cloud({k: v[o:o + n] for k, v in scn.items()}, 4096, 99, env_offset=o)
        np.testing.assert_array_equal(a.cpu().numpy().view(np.uint16), assign)
        np.testing.assert_array_equal(lab.cpu().numpy(), labels)
        np.testing.assert_allclose(got[..., :3].cpu().numpy(), pts, rtol=0, atol=2e-6)


def test_problem_batch_rows_do_not_depend_on_the_shard():
    from mpinets_amd.scenes import make_problem_batch

    dev = torch.device("cuda:0")
    kw = dict(seed=8, device=dev, kinds=("tabletop", "cubby", "dresser"), M1=40, M2=16)
    for pool, clouds in ((None, False), (5, True), (None, True)):
        whole = make_problem_batch(14, scene_pool=pool, device_clouds=clouds, **kw)
        for o, n in ((0, 6), (6, 8)):
            part = make_problem_batch(n, scene_pool=pool, device_clouds=clouds, env_offset=o, total_envs=14, **kw)
            for k, v in whole.items():
                if torch.is_tensor(v) and v.size(0) == 14:
                    assert torch.equal(part[k], v[o:o + n]), (k, pool, clouds, o)


def test_bench_distributed_path_on_a_real_rccl_communicator():
    """`bench.py`'s N > 1 code path -- shard.init -> init_process_group("nccl", device_id=...), barrier, MAX all-reduce,
    gather_to_rank0, all_gather_object -- executed on a REAL RCCL communicator: one rank under torch.distributed.run with
    the process group forced for world size 1 (MPX_DIST_FORCE=1).  The box has one GPU, so this is every RCCL call of the
    8-GPU run except the inter-GPU transport itself.  The JSON must say so itself (rccl_ranks, per-rank records).
    Reference launcher: mpinets/run_training.py:71-77 (one process per GPU under DDP)."""
    import json

    root = os.path.dirname(HERE)
    env = dict(os.environ, MPX_DIST_FORCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("MPX_DIST_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "1", "--envs", "1040", "--steps", "2",
           "--warmup", "1", "--extra", "0", "--fast-steps", "1", "--cpu-envs", "0", "--scene-pool", "64"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["dist_backend"] == "nccl" and out["rccl_ranks"] == 1 and out["n_gpus"] == 1
    rec = out["ranks"]
    assert len(rec) == 1 and rec[0]["rank"] == 0 and rec[0]["device"] == "cuda:0" and rec[0]["env_ids"] == [0, 1040]
    assert "MI3" in rec[0]["device_name"] or "gfx" in rec[0]["gcn_arch"], rec
    assert abs(rec[0]["ms_per_step"] - out["ms_per_step"]) <= 0.05 * out["ms_per_step"] + 0.5  # MAX over one rank
    assert out["value"] > 0 and out["result_check"]["gathered_q"] == [1040, 7]
    assert out["fast_mode"]["value"] > 0


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_two_ranks_report_sharded_extras_and_both_scalings(scaling):
    """`python bench.py --gpus 2` starts its own two ranks (free port); on this one-GPU box they share the device over
    gloo (MPX_SHARE_GPU=1, the JSON says so).  Checks the N-rank fields of the line: config 3's trajectories split over
    the ranks with per-rank times, config 1 on every rank, the data-parallel training step, strong / weak accounting.
    BASELINE configs[3] ("sharded"), model.py:293-314."""
    import json

    root = os.path.dirname(HERE)
    env = dict(os.environ, MPX_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("MPX_DIST_BACKEND", "MPX_DIST_FORCE", "WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--scaling", scaling, "--envs", "520", "--global-envs",
           "1041", "--steps", "1", "--warmup", "1", "--fast-steps", "0", "--static-steps", "0",
           "--all-slots-steps", "0", "--train-steps", "1", "--cpu-envs", "0", "--scene-pool", "32"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["scaling"] == scaling and out["dist_backend"] == "gloo" and out["rccl_ranks"] == 0
    assert out["config"]["devices_shared"] is True
    ids = [rec["env_ids"] for rec in out["ranks"]]
    if scaling == "weak":
        assert ids == [[0, 520], [520, 1040]] and out["config"]["global_envs"] == 1040
    else:  # 1041 environments over two ranks: 521 + 520
        assert ids == [[0, 521], [521, 1041]] and out["config"]["global_envs"] == 1041
    assert abs(out["value"] - out["config"]["global_envs"] * out["steps"] / (out["ms_per_step"] * out["steps"] * 1e-3)) < 1e-6 * out["value"]
    assert out["result_check"]["gathered_q"] == [out["config"]["global_envs"], 7]
    assert "frac_nominal" in out["roofline"] and out["roofline"]["frac_nominal"] > out["roofline"]["frac"] > 0
    c4 = out["extra_configs"]["c4_collision_validation"]
    assert len(c4["ms_per_rank"]) == 2 and c4["envs"] == sum(c4["envs_per_rank"]) and min(c4["envs_per_rank"]) >= 520
    assert c4["ms"] == max(c4["ms_per_rank"]) and abs(c4["env_waypoints_per_s"] - c4["envs"] * 50 / c4["ms"] * 1e3) < 1e-6 * c4["env_waypoints_per_s"]
    assert 0.0 <= c4["collision_rate"] <= 1.0
    c2 = out["extra_configs"]["c2_fk_sdf_1024"]
    assert len(c2["ms_per_rank"]) == 2 and c2["envs"] == sum(c2["envs_per_rank"])
    tr = out["extra_configs"]["n1_training_step"]
    assert tr["allreduce_ranks"] == 2 and tr["batch_10"]["samples_per_gpu"] == 10 and tr["batch_10"]["samples_per_s"] > 0
    assert np.isfinite(tr["batch_256"]["loss"])
    assert out["extra_configs"]["c1_single_problem"]["ms_per_step"] > 0  # (rank 0 alone, after the others were released)


@pytest.mark.parametrize("launcher,scaling", [("driver", "weak"), ("self", "strong")])
def test_bench_eight_ranks_dress_rehearsal(launcher, scaling):
    """The 1 -> 8 GPU curve is taken by the driver in ONE run on a node no round has touched, so the 8-rank command is
    rehearsed here in full: eight processes (sharing this box's GPU over gloo: MPX_SHARE_GPU / MPX_SHARED_DEVICES, flagged
    in the JSON), every default-on extra, small shares.  `driver` is the driver's own command line
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P bench.py --gpus 8
    --steps K --warmup W`), `self` lets bench.py start its ranks.  Checks: 8 rank records with contiguous environment
    ranges, the gathered result's shape, config 3 split 8 ways, the data-parallel training step over 8 ranks, rank 0's
    single-GPU extras after the others were released, and a wall-time bound.  north_star: "no RCCL collectives on the
    step, only a final host gather"; reference launcher: mpinets/run_training.py:71-77."""
    import json
    import time

    root = os.path.dirname(HERE)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("MPX_DIST_BACKEND", "MPX_DIST_FORCE", "WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MPX_SHARED_DEVICES"):
        env.pop(k, None)
    world, per = 8, 64
    total = 515 if scaling == "strong" else world * per  # strong: 3 shares of 65 + 5 of 64 (padded gather)
    small = ["--envs", str(per), "--global-envs", str(total), "--scaling", scaling, "--steps", "2", "--warmup", "1",
             "--fast-steps", "1", "--static-steps", "1", "--all-slots-steps", "1", "--train-steps", "1",
             "--cpu-envs", "0", "--scene-pool", "16"]
    if launcher == "driver":
        env.update(MPX_SHARED_DEVICES="1", MPX_DIST_BACKEND="gloo")  # (what bench.py sets for its own ranks under MPX_SHARE_GPU=1)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
               "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", str(world)] + small
    else:
        env["MPX_SHARE_GPU"] = "1"
        cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world)] + small
    t0 = time.time()
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    wall = time.time() - t0
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly ONE JSON line (rank 0)"
    out = json.loads(lines[0])
    assert out["n_gpus"] == world and out["scaling"] == scaling and out["dist_backend"] == "gloo" and out["rccl_ranks"] == 0
    assert out["config"]["devices_shared"] is True and out["config"]["global_envs"] == total
    recs = sorted(out["ranks"], key=lambda x: x["rank"])
    assert [x["rank"] for x in recs] == list(range(world)) and [x["local_rank"] for x in recs] == list(range(world))
    ids = [x["env_ids"] for x in recs]
    assert ids[0][0] == 0 and ids[-1][1] == total and all(a[1] == b[0] for a, b in zip(ids, ids[1:]))
    sizes = [b - a for a, b in ids]
    assert sizes == ([65] * 3 + [64] * 5 if scaling == "strong" else [per] * world)
    assert out["result_check"]["gathered_q"] == [total, 7]
    assert out["ms_per_step"] >= max(x["ms_per_step"] for x in recs) * 0.999  # MAX over ranks
    assert abs(out["value"] - total / (out["ms_per_step"] * 1e-3)) < 1e-6 * out["value"]
    c4 = out["extra_configs"]["c4_collision_validation"]
    assert len(c4["ms_per_rank"]) == world and len(c4["envs_per_rank"]) == world and c4["ms"] == max(c4["ms_per_rank"])
    assert len(out["extra_configs"]["c2_fk_sdf_1024"]["ms_per_rank"]) == world
    tr = out["extra_configs"]["n1_training_step"]
    assert tr["allreduce_ranks"] == world and np.isfinite(tr["batch_10"]["loss"]) and np.isfinite(tr["batch_256"]["loss"])
    assert out["fast_mode"]["value"] > 0
    assert out["extra_configs"]["c1_single_problem"]["ms_per_step"] > 0  # rank 0 alone, after the release
    assert wall < 900, f"8-rank rehearsal took {wall:.0f} s"


def test_shard_init_takes_a_free_port_for_a_forced_single_rank(monkeypatch):
    from mpinets_amd import shard

    monkeypatch.delenv("MASTER_PORT", raising=False)
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    with pytest.raises(RuntimeError, match="MASTER_PORT is not set"):
        shard.init(backend="gloo")


def test_shard_init_device_defaults(monkeypatch):
    """shard.init(): a gloo rank whose LOCAL_RANK exceeds the device count shares a GPU (device = LOCAL_RANK mod count);
    an RCCL rank in the same position fails with a clear message instead of 'invalid device ordinal'."""
    from mpinets_amd import shard

    ndev = torch.cuda.device_count()
    monkeypatch.setenv("LOCAL_RANK", str(ndev))  # one past the last device
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.delenv("MPX_DIST_FORCE", raising=False)
    prev = torch.cuda.current_device()
    try:
        shard.init(backend="gloo")
        assert torch.cuda.current_device() == 0
        with pytest.raises(RuntimeError, match="RCCL needs one device per rank"):
            shard.init(backend="nccl")
    finally:
        torch.cuda.set_device(prev)
