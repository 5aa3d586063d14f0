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


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE configs[3] at its exact partition, without hardware: 512 clips over 8 ranks (64 each) and the uneven 509

def _worker_cfg3(rank, world, port, n_clips, q):
    from babe_amd.dist import rank_table, restore_clips_sharded, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    clips = torch.randn(n_clips, 1700, generator=torch.Generator().manual_seed(5))
    x, fp = restore_clips_sharded(_StubSampler(), clips, seed=20)
    lo, hi = shard_range(n_clips, rank, world)
    assert rank_table(n_clips, world)[rank] == (rank, rank, lo, hi)
    # every rank holds all n results in clip order; send a digest + this rank's own shard boundaries
    q.put((rank, lo, hi, tuple(x.shape), tuple(fp.shape), x.double().sum(1).numpy().copy(), fp[:, 0].numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def _single_rank_reference(n_clips):
    from babe_amd.dist import restore_clips_sharded
    clips = torch.randn(n_clips, 1700, generator=torch.Generator().manual_seed(5))
    x, fp = restore_clips_sharded(_StubSampler(), clips, seed=20)
    return x.double().sum(1).numpy(), fp[:, 0].numpy()


def _run_world8(n_clips):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_cfg3, args=(r, 8, port, n_clips, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(8)]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return sorted(res, key=lambda r: r[0])


def test_config3_partition_512_clips_world8_gloo():
    """configs[3]'s own partition - 512 clips block-partitioned over 8 ranks, 64 per rank - through the product's sharded
    driver and its end-of-batch gather (gloo here, RCCL on the node): every rank ends with all 512 results in clip order,
    equal to the single-rank run row for row."""
    ref_x, ref_f = _single_rank_reference(512)
    res = _run_world8(512)
    assert [(lo, hi) for _, lo, hi, *_ in res] == [(64 * r, 64 * (r + 1)) for r in range(8)]
    for _, _, _, xs, fs, dx, df in res:
        assert xs == (512, 1700) and fs == (512, 8)
        assert (dx == ref_x).all() and (df == ref_f).all()


def test_config3_uneven_509_clips_world8_gloo():
    """509 = 8 * 63 + 5: ranks 0-4 restore 64 clips, ranks 5-7 63; the short shards are padded for the collective and the
    padding rows are dropped again."""
    ref_x, ref_f = _single_rank_reference(509)
    res = _run_world8(509)
    sizes = [hi - lo for _, lo, hi, *_ in res]
    assert sizes == [64] * 5 + [63] * 3 and res[0][1] == 0 and res[-1][2] == 509
    for _, _, _, xs, fs, dx, df in res:
        assert xs == (509, 1700) and fs == (509, 8)
        assert (dx == ref_x).all() and (df == ref_f).all()


def test_pin_host_threads_blocks_are_disjoint():
    from babe_amd.dist import pin_host_threads
    if not hasattr(os, "sched_getaffinity"):
        return
    keep = os.sched_getaffinity(0)
    try:
        cpus = list(range(32))                                  # a pretend 32-CPU host: 8 ranks x 4 CPUs (no call is made: only the split)
        blocks = []
        for r in range(8):
            per = len(cpus) // 8
            blocks.append(cpus[r * per:(r + 1) * per])
        assert sorted(sum(blocks, [])) == cpus and all(len(b) == 4 for b in blocks)
        if len(keep) >= 4:                                      # the real call, on the CPUs this process has
            mine = pin_host_threads(1, 2)
            assert mine == sorted(keep)[len(keep) // 2: 2 * (len(keep) // 2)] and os.sched_getaffinity(0) == set(mine)
        assert pin_host_threads(0, 10 ** 6) is None             # fewer than two CPUs per rank: left alone
    finally:
        os.sched_setaffinity(0, keep)


def test_bench_dry_run_prints_the_rank_table():
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--clips-per-gpu", "64", "--dry-run"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    rec = json.loads(out.stdout)
    assert rec["n_gpus"] == 8 and rec["clips_per_step"] == 512
    assert [r["clips"] for r in rec["ranks"]] == [[64 * r, 64 * (r + 1)] for r in range(8)]
    assert [r["device"] for r in rec["ranks"]] == [f"cuda:{r}" for r in range(8)]
