"""bench.py itself on the GPU: the N > 1 self-check through its one-GPU rehearsal and through a real 1-rank RCCL
communicator under the launcher — the code the driver's multi-GPU run goes through must not meet its first execution there."""
import json
import os
import socket
import subprocess
import sys
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_json(out):
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and lines, out.stdout[-2000:] + out.stderr[-3000:]
    return json.loads(lines[-1])


def test_loopback_rehearsal_checks_every_rank(gpu):
    """`bench.py --loopback 4`: four logical ranks of one process (dvs_comm_create_loopback), 16 frame pairs per step split over them; every
    rank verifies the match job that depends on the exchange against the oracle, rank 0's line carries the verdicts"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--loopback", "4", "--global-batch", "16", "--steps", "4", "--warmup", "1"],
                         capture_output=True, text=True, env=env, timeout=600)
    d = _last_json(out)
    assert d["metric"].startswith("DIAGNOSTIC") and d["logical_ranks"] == 4 and not d["errors"]
    bc = d["rccl"]["boundary_check"]
    assert bc["ranks_checked"] == 4 and bc["result"] == "identical to the oracle on every rank", bc
    assert [v["predecessor"]["rank"] for v in bc["per_rank"]] == [3, 0, 1, 2]
    assert bc["per_rank"][0]["predecessor"]["batch"] == bc["per_rank"][0]["batch"] - 1       # rank 0 wraps to the call before
    assert d["rccl"]["allgather_bytes"] == 4 * 64832


def test_one_rank_under_the_launcher_runs_real_rccl(gpu):
    """the driver's form (`python -m torch.distributed.run ... bench.py --gpus 1`): an RCCL communicator of one rank, the exchange in the loop,
    the boundary check against the oracle in the result line"""
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = str(so.getsockname()[1])
    env = dict({k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                          "--master-port", port, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--batch", "8", "--steps", "6", "--warmup", "2",
                          "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=900)
    d = _last_json(out)
    assert d["n_gpus"] == 1 and d["value"] > 0 and "diagnostic" not in d
    r = d["rccl"]
    assert r["nranks"] == 1 and r["version"] > 0 and r["boundary_check"]["result"] == "identical to the oracle on every rank", r
    assert r["boundary_check"]["per_rank"][0]["predecessor"] == {"rank": 0, "batch": r["boundary_check"]["per_rank"][0]["batch"] - 1}
    assert r["allgather_us"] is not None and r["allgather_bytes"] == 64832
