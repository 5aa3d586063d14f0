"""The N > 1 path with REAL sampler outputs on the one GPU a test box has: two ranks (gloo rendezvous, both on GPU
0) restore 3 clips - uneven shards 2 + 1 - through babe_amd.dist.restore_clips_sharded and the gathered [n, L] must equal a
single-rank run row for row; and bench.py's own launch path (--gpus 2 without a launcher: spawn_ranks -> torch.distributed.run
-> one process per rank -> end-of-step gather) must produce its JSON line.  Ranks are child processes started before they
touch the GPU.  Needs a MI355X.  (An 8-GPU RCCL run is the driver's to make: (e) stays 'unmeasured on hardware' until a SCALE
record exists.)"""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _env():
    e = dict(os.environ)
    e["BABE_DIST_BACKEND"] = "gloo"
    e["MASTER_ADDR"] = "127.0.0.1"
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return e


def test_two_ranks_uneven_shards_equal_single_rank_row_for_row(tmp_path):
    script = os.path.join(ROOT, "tools", "dist_restore_check.py")
    one, two = str(tmp_path / "w1.pt"), str(tmp_path / "w2.pt")
    r1 = subprocess.run([sys.executable, script, one], env=_env(), capture_output=True, text=True, timeout=900)
    assert r1.returncode == 0, r1.stderr[-2000:]
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                         "127.0.0.1", "--master-port", str(_free_port()), script, two],
                        env=_env(), capture_output=True, text=True, timeout=900)
    assert r2.returncode == 0, r2.stderr[-2000:]
    a, b = torch.load(one), torch.load(two)
    assert a["world"] == 1 and b["world"] == 2
    assert a["x"].shape == b["x"].shape == (3, 110000) and a["fp"].shape == b["fp"].shape
    assert bool(torch.isfinite(b["x"]).all()) and float(b["x"].std()) > 0
    # same kernels, same per-clip seeds, same order inside a clip: bit-identical rows whatever the sharding
    assert torch.equal(a["x"], b["x"]), float((a["x"] - b["x"]).abs().max())
    assert torch.equal(a["fp"], b["fp"])


def test_bench_two_ranks_shared_gpu_prints_its_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--T", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline", "--profile-steps", "0"], env=_env(), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["output_finite"] is True and d["value"] > 0


def _env_nccl():
    e = dict(os.environ)
    e.pop("BABE_DIST_BACKEND", None)                     # default backend = nccl (RCCL on ROCm)
    e["MASTER_ADDR"] = "127.0.0.1"
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return e


_NCCL_W1 = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from babe_amd.dist import gather_results
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"
g = torch.Generator().manual_seed(3)
x = torch.randn(3, 4099, generator=g).to(dev)
fp = torch.randn(3, 20, generator=g).to(dev)
# without the flag a world of one returns its inputs untouched; with it the padded buffer goes through all_gather_into_tensor
x0, fp0 = gather_results(x, fp)
assert x0 is x and fp0 is fp
x1, fp1 = gather_results(x, fp, force_collective=True)
torch.cuda.synchronize()
assert x1 is not x and x1.shape == x.shape and fp1.shape == fp.shape
assert torch.equal(x1, x) and torch.equal(fp1, fp)
# an empty shard (a rank without clips) still takes part
xe, fpe = gather_results(x[:0], fp[:0], force_collective=True)
assert xe.shape == (0, 4099) and fpe.shape == (0, 20)
dist.barrier()
dist.destroy_process_group()
print("NCCL_W1_OK")
"""


def test_rccl_branch_of_gather_results_world_size_one():
    """The `nccl` branch of gather_results (init_process_group("nccl", device_id=...), all_gather of the shard sizes,
    all_gather_into_tensor on the padded buffer) executed on the one GPU a test box has: a world of ONE rank with the
    early return bypassed.  (Two RCCL ranks on one device are refused by RCCL; the 8-GPU job is the driver's.)"""
    e = _env_nccl()
    e["MASTER_PORT"] = str(_free_port())
    r = subprocess.run([sys.executable, "-c", _NCCL_W1, ROOT], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "NCCL_W1_OK" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])


def test_bench_one_rank_under_torchrun_takes_the_rccl_path():
    """`torchrun --nproc-per-node 1 bench.py --gpus 1`: the launch path of the driver's N > 1 runs with one rank - process
    group on RCCL, barrier, end-of-step all_gather_into_tensor, max-over-ranks all_reduce - and the line says so."""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr",
                        "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1",
                        "--T", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--profile-steps", "0"],
                       env=_env_nccl(), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["output_finite"] is True and d["value"] > 0
    assert "RCCL all_gather" in d["config"]["parallelism"], d["config"]["parallelism"]
