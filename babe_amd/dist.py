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


def gather_results(x_local, fp_local, n_total=None):
    """x_local [b,L], fp_local [b,P] -> (x_all [n,L], fp_all [n,P]) on every rank, in global clip order.
    Shards may differ by one clip; they are padded to the largest shard for the collective."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
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
