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
