"""Child program of tests/test_gpu_dist.py: restore 3 short clips with babe_amd.dist.restore_clips_sharded on however many
ranks the launcher started (WORLD_SIZE; gloo when BABE_DIST_BACKEND=gloo, all ranks may share GPU 0) and save the gathered
result of rank 0.  Reduced width, T = 2, 22.05 kHz."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(out_path, n_clips=3):
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dev_idx = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_idx)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("BABE_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_idx))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    from babe_amd.config import default_args
    from babe_amd.diff_params.edm import EDM
    from babe_amd.dist import restore_clips_sharded
    from babe_amd.networks.cqtdiff_plus import Unet_CQT_oct_with_attention, init_state_dict
    from babe_amd.testing.blind_bwe_sampler import BlindSampler
    fs, segL = 22050, 92092
    Ns = [8, 8, 8, 8, 16, 16, 16]
    args = default_args(sample_rate=fs, audio_len=segL, Ns=Ns, T=2, start_sigma=0.05)
    args.tester.blind_bwe.optimization.mu = [100.0, 1.0]
    net = Unet_CQT_oct_with_attention(args, "cuda")
    net.load_state_dict(init_state_dict(Ns, args.network.num_dils, seed=4, gate_scale=1.0))
    smp = BlindSampler(net, EDM(args), args, batch_semantics="per_clip", noise_device="cuda")
    Lc = 110000                                          # 1.2 segments: two segments per clip, the second zero-padded
    g = torch.Generator().manual_seed(99)
    t_ax = torch.arange(Lc) / fs
    clips = torch.stack([sum(0.05 / (k + 1) * torch.sin(2 * torch.pi * (180.0 + 40 * c) * (k + 1) * t_ax) for k in range(8))
                         + 0.02 * torch.randn(Lc, generator=g) for c in range(n_clips)]).cuda()
    x, fp = restore_clips_sharded(smp, clips, seed=500)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    if rank == 0:
        torch.save({"x": x.cpu(), "fp": fp.cpu(), "world": world}, out_path)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
