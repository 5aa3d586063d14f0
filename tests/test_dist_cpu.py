"""world_size-2 gloo test of the clip sharding + end-of-batch gather (the N>1 path of bench.py)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_clips, q):
    from babe_amd.dist import gather_results, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(n_clips, rank, world)
    x = torch.stack([torch.full((16,), float(c)) for c in range(lo, hi)]) if hi > lo else torch.zeros(0, 16)
    fp = torch.stack([torch.arange(10.0) + 100 * c for c in range(lo, hi)]) if hi > lo else torch.zeros(0, 10)
    xa, fa = gather_results(x, fp)
    q.put((rank, xa[:, 0].tolist(), fa[:, 0].tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions():
    from babe_amd.dist import shard_range
    for n in (0, 1, 5, 8, 513):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_gather_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    n_clips = 5
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_clips, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, xs, fs in res:
        assert xs == [float(c) for c in range(n_clips)]
        assert fs == [100.0 * c for c in range(n_clips)]


class _StubSampler:
    """predict_blind_bwe stand-in for the host-logic test: 'restores' a segment by adding 1 and reports its mean as filter."""

    def __init__(self):
        import types
        ic = types.SimpleNamespace(fc=[1.0, 2.0])
        self.args = types.SimpleNamespace(exp=types.SimpleNamespace(audio_len=1000),
                                          tester=types.SimpleNamespace(blind_bwe=types.SimpleNamespace(initial_conditions=ic)))

    def predict_blind_bwe(self, y):
        noise = torch.randn(y.shape)                      # (seeded per clip by restore_clips_sharded)
        return y + 1.0 + noise, torch.stack([y.mean(1, keepdim=True).expand(-1, 2), y.std(1, keepdim=True).expand(-1, 2)], 1)


def _worker_restore(rank, world, port, q):
    from babe_amd.dist import restore_clips_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    clips = torch.randn(3, 1700, generator=torch.Generator().manual_seed(3))
    x, fp = restore_clips_sharded(_StubSampler(), clips, seed=10)
    q.put((rank, x.numpy().copy(), fp.numpy().copy()))          # (by value: the producer exits before the consumer reads)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_restore_clips_sharded_world2_equals_world1():
    """3 clips on 2 ranks (shards of 2 and 1) through the product's sharded driver: segmentation, per-clip seeding, padding
    of the uneven shard and the gather give every rank the single-rank result, row for row."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p0 = ctx.Process(target=_worker_restore, args=(0, 1, _free_port(), q))
    p0.start()
    _, x1, f1 = q.get(timeout=120)
    p0.join(60)
    port = _free_port()
    procs = [ctx.Process(target=_worker_restore, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert x1.shape == (3, 1700) and f1.shape == (3, 8)
    for _, x2, f2 in res:
        assert (x2 == x1).all() and (f2 == f1).all()


def _worker_w1(port, q):
    from babe_amd.dist import gather_results
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    x, fp = torch.arange(48.0).reshape(3, 16), torch.arange(30.0).reshape(3, 10)
    x0, _ = gather_results(x, fp)                                   # world of one: inputs returned untouched
    x1, f1 = gather_results(x, fp, force_collective=True)           # ... unless the collective is forced (the RCCL smoke's switch)
    q.put((x0 is x, x1 is x, bool(torch.equal(x1, x) and torch.equal(f1, fp))))
    dist.destroy_process_group()


def test_gather_world1_force_collective():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_w1, args=(_free_port(), q))
    p.start()
    same0, same1, equal = q.get(timeout=120)
    p.join(60)
    assert p.exitcode == 0 and same0 and not same1 and equal
