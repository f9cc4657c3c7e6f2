"""N > 1 path on CPU: env sharding + barrier + max-over-ranks + final gather with gloo, world_size 2 and 8 (the
driver's scaling run uses 8 ranks: the same collectives, uneven shares and padded gathers at that width)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, envs_per_rank, out_dir, strong_total=0):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from mpinets_amd import shard

    r, w, _ = shard.init(backend="gloo")
    assert (r, w) == (rank, world)
    # weak scaling: equal shares; strong scaling: split_even shares of a fixed total, which may differ by one
    ids = shard.split_even(strong_total, w)[r] if strong_total else shard.env_range(r, w, envs_per_rank)
    # stand-in for the per-rank rollout result: a deterministic function of the GLOBAL env id
    q = torch.tensor([[float(i) + 0.1 * j for j in range(7)] for i in ids])
    flags = torch.tensor([i % 3 == 0 for i in ids], dtype=torch.int32)
    shard.barrier()
    t = shard.max_over_ranks(1.0 + rank)
    assert t == float(world)
    q_all = shard.gather_to_rank0(q)
    f_all = shard.gather_to_rank0(flags)
    if r == 0:
        np.save(os.path.join(out_dir, "q.npy"), q_all.numpy())
        np.save(os.path.join(out_dir, "f.npy"), f_all.numpy())
    else:
        assert q_all is None and f_all is None
    shard.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_shard_equals_single_rank(tmp_path):
    world, per_rank = 2, 5
    mp.spawn(_worker, args=(world, _free_port(), per_rank, str(tmp_path)), nprocs=world, join=True)
    q = np.load(tmp_path / "q.npy")
    f = np.load(tmp_path / "f.npy")
    ids = np.arange(world * per_rank)
    np.testing.assert_allclose(q, ids[:, None] + 0.1 * np.arange(7)[None], rtol=1e-6)
    np.testing.assert_array_equal(f, (ids % 3 == 0).astype(np.int32))


def test_strong_scaling_shares_of_unequal_size_gather_in_order(tmp_path):
    """bench.py --scaling strong: 11 environments over 2 ranks = 6 + 5; the final gather pads and trims."""
    world, total = 2, 11
    mp.spawn(_worker, args=(world, _free_port(), 0, str(tmp_path), total), nprocs=world, join=True)
    q, f = np.load(tmp_path / "q.npy"), np.load(tmp_path / "f.npy")
    ids = np.arange(total)
    assert q.shape == (total, 7) and f.shape == (total,)
    np.testing.assert_allclose(q, ids[:, None] + 0.1 * np.arange(7)[None], rtol=1e-6)
    np.testing.assert_array_equal(f, (ids % 3 == 0).astype(np.int32))


def _scene_worker(rank, world, port, per_rank, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from mpinets_amd import shard
    from mpinets_amd.scenes import make_scenes
    from oracle import oracle as orc

    orc.build()
    r, w, _ = shard.init(backend="gloo")
    ids = shard.env_range(r, w, per_rank)
    scn = make_scenes(world * per_rank, 3, ("tabletop", "cubby", "dresser"), 40, 16)  # same seed on every rank
    mine = {k: v[ids.start:ids.stop] for k, v in scn.items()}
    pts, assign, labels, _ = orc.scene_cloud(mine, 512, 77, env_offset=ids.start)  # keyed by the GLOBAL env id
    p_all = shard.gather_to_rank0(torch.from_numpy(pts))
    a_all = shard.gather_to_rank0(torch.from_numpy(assign.astype(np.int32)))
    if r == 0:
        np.save(os.path.join(out_dir, "p.npy"), p_all.numpy())
        np.save(os.path.join(out_dir, "a.npy"), a_all.numpy())
    shard.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("total", [8 * 5, 67, 8])
def test_eight_ranks_gather_in_order(tmp_path, total):
    """The driver's 8-rank run: weak shares (8 x 5), an uneven strong split (67 = 3 x 9 + 5 x 8: the final gather pads to
    9 rows and trims) and one environment per rank.  north_star: "no RCCL collectives on the step, only a final host
    gather"; run_training.py:71-77 is the reference's one-process-per-GPU launcher."""
    world = 8
    mp.spawn(_worker, args=(world, _free_port(), 5, str(tmp_path), 0 if total == 40 else total), nprocs=world, join=True)
    q, f = np.load(tmp_path / "q.npy"), np.load(tmp_path / "f.npy")
    ids = np.arange(total)
    assert q.shape == (total, 7) and f.shape == (total,)
    np.testing.assert_allclose(q, ids[:, None] + 0.1 * np.arange(7)[None], rtol=1e-6)
    np.testing.assert_array_equal(f, (ids % 3 == 0).astype(np.int32))


def test_sharded_scene_draw_equals_unsharded(tmp_path, oracle):
    """The per-environment random draws of the step are keyed by the global environment id (env_offset of
    mpx_scene_cloud, restated by the oracle): two ranks' gathered draws == one process's draw, bit for bit.
    (The same statement on the real engine, on the GPU: tests/test_gpu_shard.py.)"""
    from mpinets_amd.scenes import make_scenes

    world, per_rank = 2, 3
    mp.spawn(_scene_worker, args=(world, _free_port(), per_rank, str(tmp_path)), nprocs=world, join=True)
    scn = make_scenes(world * per_rank, 3, ("tabletop", "cubby", "dresser"), 40, 16)
    pts, assign, _, _ = oracle.scene_cloud(scn, 512, 77)
    np.testing.assert_array_equal(np.load(tmp_path / "p.npy"), pts)
    np.testing.assert_array_equal(np.load(tmp_path / "a.npy"), assign.astype(np.int32))
    # and the offset matters: the second shard drawn with offset 0 differs
    other, _, _, _ = oracle.scene_cloud({k: v[per_rank:] for k, v in scn.items()}, 512, 77)
    assert not np.array_equal(other, pts[per_rank:])


def test_split_even_covers_everything():
    from mpinets_amd.shard import env_range, split_even

    for total, w in [(8192, 8), (10, 3), (7, 8), (65536, 8)]:
        parts = split_even(total, w)
        assert sum(len(p) for p in parts) == total and parts[0].start == 0 and parts[-1].stop == total
        assert all(a.stop == b.start for a, b in zip(parts, parts[1:]))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    assert list(env_range(3, 8, 1024))[:2] == [3072, 3073]


def _grad_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from mpinets_amd import shard

    shard.init(backend="gloo")
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(7, 33), torch.nn.LeakyReLU(), torch.nn.Linear(33, 5))
    x = torch.arange(4 * 7, dtype=torch.float32).reshape(4, 7) / 10 + rank  # a different batch per rank
    net(x).square().mean().backward()
    calls = shard.allreduce_gradients(list(net.parameters()), bucket_bytes=512)  # small buckets: several collectives
    assert calls >= 2
    if rank == 0:
        torch.save([p.grad.clone() for p in net.parameters()], os.path.join(out_dir, "g.pt"))
    shard.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_gradient_allreduce_averages_over_ranks(tmp_path, world):
    """Row N1: bucketed gradient all-reduce == the gradient of the mean loss over all ranks' batches."""
    mp.spawn(_grad_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = torch.load(tmp_path / "g.pt")
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(7, 33), torch.nn.LeakyReLU(), torch.nn.Linear(33, 5))
    loss = 0
    for rank in range(world):
        x = torch.arange(4 * 7, dtype=torch.float32).reshape(4, 7) / 10 + rank
        loss = loss + net(x).square().mean() / world
    loss.backward()
    for g, p in zip(got, net.parameters()):
        torch.testing.assert_close(g, p.grad, rtol=1e-5, atol=1e-6)
