"""C++ adapter headers (include/dvslam/*.hpp) and the multi-rank exchange step.
CPU: the adapters compile against the C-ABI with plain g++, and the boundary-descriptor exchange of
dvslam_amd/dist.py is exercised with 2 gloo ranks.  GPU: the compiled adapter program runs end to end."""
import os
import subprocess
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "dynamic-visual-slam_amd", "lib")


def _build_adapter_smoke(tmpdir, hiplib):
    exe = os.path.join(str(tmpdir), "adapter_smoke")
    cmd = ["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "adapter_smoke.cpp"),
           "-o", exe, "-L" + LIBDIR, "-ldvslam_hip", "-Wl,-rpath," + LIBDIR, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return exe


def test_adapters_compile_and_refuse_without_gpu(tmp_path, hiplib):
    from dvslam_amd import device_count
    exe = _build_adapter_smoke(tmp_path, hiplib)
    rc = subprocess.call([exe])
    assert rc == (0 if device_count() > 0 else 3)


@pytest.mark.gpu
def test_adapters_run_on_gpu(tmp_path, gpu, hiplib):
    exe = _build_adapter_smoke(tmp_path, hiplib)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "BA success=1" in out.stdout


_WORKER = r"""
import os, sys
sys.path.insert(0, os.path.join(sys.argv[1], "dynamic-visual-slam_amd"))
import torch, torch.distributed as dist
from dvslam_amd import dist as dvdist
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
cap = 2024
assert list(dvdist.shard_range(world, rank, 8)) == list(range(rank * 8, rank * 8 + 8))
desc = torch.full((cap, 32), 10 + rank, dtype=torch.uint8)
desc[5, 7] = 200 + rank
n = torch.tensor(1900 + rank, dtype=torch.int32)
d, m = dvdist.exchange_boundary(desc, n, cap)
prev = (rank - 1) % world
assert int(m) == 1900 + prev, (rank, int(m))
assert int(d[0, 0]) == 10 + prev and int(d[5, 7]) == 200 + prev and d.shape == (cap, 32)
# second round with different payloads: no stale data
d2, m2 = dvdist.exchange_boundary(desc + 1, n + 7, cap)
assert int(m2) == 1907 + prev and int(d2[0, 0]) == 11 + prev
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_boundary_exchange_two_gloo_ranks(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29631", str(script), ROOT], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.count("ok") == 2


def test_single_rank_exchange_is_identity():
    import torch
    from dvslam_amd import dist as dvdist
    desc = torch.arange(64 * 32, dtype=torch.int64).remainder(251).to(torch.uint8).reshape(64, 32)
    d, n = dvdist.exchange_boundary(desc, torch.tensor(17, dtype=torch.int32), 64)
    assert int(n) == 17 and torch.equal(d, desc)
