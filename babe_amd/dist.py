"""Clip-parallel data parallelism: independent clips are block-partitioned over ranks (one process per
GPU, weights replicated, no per-step communication) and the restored audio + estimated filters are
gathered once at the end of a batch with a single all_gather (RCCL over xGMI on the GPU box, gloo in
the CPU tests).  The reference itself has no distribution of any kind."""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Static block partition: rank r gets items [lo, hi)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def rank_table(n_clips, world, n_devices=None):
    """The static plan of a sharded job as rows (rank, device index, first clip, one-past-last clip): what `bench.py --gpus N
    --dry-run` prints and what restore_clips_sharded executes (rank r restores clips [lo, hi) on device r mod n_devices)."""
    nd = n_devices or world
    return [(r, r % max(nd, 1)) + shard_range(n_clips, r, world) for r in range(world)]


def pin_host_threads(local_rank, local_world, cpus=None):
    """Give each rank of a node its own contiguous block of the host's CPUs (os.sched_setaffinity, in-process, BEFORE the first
    GPU call), so that the N Python enqueue loops of an N-GPU job - ~1100 launches per score evaluation each - do not migrate
    over each other.  cpus: the CPU ids to divide (default: the ones this process may run on).  Returns the block, or None
    where the platform has no affinity call or there are fewer than two CPUs per rank (the HIP runtime's helper threads need
    a second one)."""
    import os
    if not hasattr(os, "sched_setaffinity"):
        return None
    avail = sorted(cpus if cpus is not None else os.sched_getaffinity(0))
    per = len(avail) // max(local_world, 1)
    if per < 2:
        return None
    mine = avail[local_rank * per:(local_rank + 1) * per]
    os.sched_setaffinity(0, mine)
    return mine


def gather_results(x_local, fp_local, n_total=None, force_collective=False):
    """x_local [b,L], fp_local [b,P] -> (x_all [n,L], fp_all [n,P]) on every rank, in global clip order.
    Shards may differ by one clip; they are padded to the largest shard for the collective.
    force_collective: run the collective even in a world of one rank (a one-GPU box can then execute the RCCL branch -
    `init_process_group("nccl")` + `all_gather_into_tensor` - that an 8-GPU job takes: tests/test_gpu_dist.py, and bench.py
    when torch.distributed.run launches it with one rank)."""
    if not (dist.is_available() and dist.is_initialized()):
        return x_local, fp_local
    if dist.get_world_size() == 1 and not force_collective:
        return x_local, fp_local
    world = dist.get_world_size()
    b = torch.tensor([x_local.shape[0]], device=x_local.device, dtype=torch.int64)
    counts = [torch.zeros_like(b) for _ in range(world)]
    dist.all_gather(counts, b)
    counts = [int(c) for c in counts]
    bmax = max(counts)
    L, P = x_local.shape[1], fp_local.shape[1]
    buf = torch.zeros(bmax, L + P, device=x_local.device, dtype=torch.float32)
    buf[: x_local.shape[0], :L] = x_local
    buf[: x_local.shape[0], L:] = fp_local
    if dist.get_backend() == "gloo":                    # gloo has no all_gather_into_tensor for device tensors
        parts = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(parts, buf)
        out = torch.cat(parts, 0)
    else:
        out = torch.empty(world * bmax, L + P, device=x_local.device, dtype=torch.float32)
        dist.all_gather_into_tensor(out, buf)
    rows = [out[r * bmax: r * bmax + counts[r]] for r in range(world)]
    allr = torch.cat(rows, 0)
    return allr[:, :L].contiguous(), allr[:, L:].contiguous()


def restore_clips_sharded(sampler, clips, seed=None):
    """BASELINE configs[3]: n independent clips [n, Lc] (the same tensor on every rank, on the rank's GPU) are block-partitioned
    over the ranks (shard_range; shards may differ by one clip), each rank restores its own - a clip = the segments of
    long_file.plan_segments in ONE per-clip batch through sampler.predict_blind_bwe, cross-faded by long_file.assemble, exactly
    what bench.py times - and ONE all_gather at the end hands every rank all n restored clips and filters in clip order.  No
    communication before that.  seed: the noise of clip c is drawn from generators seeded with seed + c, so the result does
    not depend on the number of ranks (row for row equal to a single-rank run: tests/test_gpu_dist.py)."""
    from .testing.long_file import assemble, cut_segments, plan_segments
    n, Lc = clips.shape
    world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    rank = dist.get_rank() if world > 1 else 0
    lo, hi = shard_range(n, rank, world)
    segL = sampler.args.exp.audio_len
    plan = plan_segments(Lc, segL)
    xs, fps = [], []
    for c in range(lo, hi):
        if seed is not None:
            torch.manual_seed(seed + c)
            torch.cuda.manual_seed(seed + c)
        x, fp = sampler.predict_blind_bwe(cut_segments(clips[c], segL, plan))
        xs.append(assemble(x, plan, Lc, segL))
        fps.append(fp.reshape(-1))
    P = 2 * len(plan) * len(sampler.args.tester.blind_bwe.initial_conditions.fc)
    x_local = torch.stack(xs) if xs else torch.zeros(0, Lc, device=clips.device)
    fp_local = torch.stack(fps) if fps else torch.zeros(0, P, device=clips.device)
    return gather_results(x_local, fp_local)
