"""C++ adapter headers (include/dvslam/*.hpp) and the multi-rank exchange step.
CPU: the adapters compile against the C-ABI with plain g++, and the PRODUCT's boundary-descriptor exchange
(dvs_exchange_boundary over a host-transport communicator, csrc/comm.hip) runs in 2 and 3 OS processes with gloo as the transport.  GPU: the compiled adapter program runs end to end."""
import json
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


def _build_opencv_typed(tmpdir, hiplib):
    """every DVSLAM_WITH_OPENCV branch + the two drop-in headers include/dynamic_visual_slam/*.hpp, against the TEST-ONLY cv:: /
    rclcpp:: stand-ins of tests/cpp/stubs (compile check + data movement; they pin nothing about OpenCV)"""
    exe = os.path.join(str(tmpdir), "adapter_opencv")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "tests", "cpp", "stubs"),
           os.path.join(ROOT, "tests", "cpp", "adapter_opencv_compile.cpp"), "-o", exe, "-L" + LIBDIR, "-ldvslam_hip", "-Wl,-rpath," + LIBDIR,
           "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return exe


def test_opencv_typed_adapters_compile(tmp_path, hiplib):
    from dvslam_amd import device_count
    exe = _build_opencv_typed(tmp_path, hiplib)
    assert subprocess.call([exe]) == (0 if device_count() > 0 else 3)


@pytest.mark.gpu
def test_opencv_typed_adapters_run_on_gpu(tmp_path, gpu, hiplib):
    """ORB_SLAM3::ORBextractor (cv::InputArray signature, mvImagePyramid), dvslam::HammingBFMatcher and the global-namespace
    SlidingWindowBA / KeyframeData / OptimizationResult written like the reference's call sites"""
    exe = _build_opencv_typed(tmp_path, hiplib)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "opencv-typed adapters ok" in out.stdout


def _build_cpp(tmpdir, src, name):
    exe = os.path.join(str(tmpdir), name)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", src),
                           "-o", exe, "-L" + LIBDIR, "-ldvslam_hip", "-Wl,-rpath," + LIBDIR, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_undistortion_ahead_of_the_pnp_stage(tmp_path, hiplib):
    """dvslam::undistortImagePoints (the solvePnPRansac adapter's handling of camera_info's D, frontend.cpp:911-921): pixels distorted with
    plumb_bob / rational coefficients return to the distortion-free ones to < 0.02 px; host code, no GPU"""
    out = subprocess.run([_build_cpp(tmp_path, "undistort_roundtrip.cpp", "undistort_roundtrip")], capture_output=True, text=True)
    assert out.returncode == 0 and "undistort round trip ok" in out.stdout, out.stdout + out.stderr


def test_sequential_association_adapter_compiles(tmp_path, hiplib):
    from dvslam_amd import device_count
    assert subprocess.call([_build_cpp(tmp_path, "association_seq.cpp", "association_seq")]) == (0 if device_count() > 0 else 3)


@pytest.mark.gpu
def test_sequential_association_matches_the_one_by_one_loop(tmp_path, gpu, hiplib):
    """backend.cpp:735-797: an association re-triangulates its landmark before the next observation is tested.  Two observations of
    one keyframe choose landmark 5; after the first moved it, the second must fall to the twin landmark — and the whole keyframe
    must come out exactly as a literal one-by-one loop over a live database gives it (include/dvslam/association.hpp)"""
    out = subprocess.run([_build_cpp(tmp_path, "association_seq.cpp", "association_seq")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "adapter vs sequential loop: 0 differences" in out.stdout and "obs1 -> 210 (snapshot 5)" in out.stdout


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
import numpy as np
import torch, torch.distributed as dist
from dvslam_amd import dist as dvdist
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
cap = 2024
assert list(dvdist.shard_range(world, rank, 8)) == list(range(rank * 8, rank * 8 + 8))
# the PRODUCT's exchange step (dvs_exchange_boundary in csrc/comm.hip, host-transport communicator): gloo only moves the bytes
comm = dvdist.HostComm(rank, world, dvdist.gloo_all_gather())
desc = np.full((cap, 32), 10 + rank, np.uint8)
desc[5, 7] = 200 + rank
n = 1900 + rank
# one call per global batch with this rank's LAST frame; the result is the frame before this rank's FIRST frame in the global order
d, m = comm.exchange_boundary(desc, n)
if rank == 0:
    assert d is None and m is None                     # the sequence starts here
else:
    assert m == 1900 + rank - 1 and int(d[0, 0]) == 10 + rank - 1 and int(d[5, 7]) == 200 + rank - 1 and d.shape == (cap, 32)
# second batch with different payloads: rank r >= 1 sees rank r - 1's block of THIS batch, rank 0 the last rank's of the FIRST batch
d2, m2 = comm.exchange_boundary(desc + 1, n + 7)
if rank == 0:
    assert m2 == 1900 + world - 1 and int(d2[0, 0]) == 10 + world - 1 and int(d2[5, 7]) == 200 + world - 1
else:
    assert m2 == 1907 + rank - 1 and int(d2[0, 0]) == 11 + rank - 1
# third batch: rank 0 now sees the second batch's last block; the second call's result is still intact (three buffers in rotation)
d3, m3 = comm.exchange_boundary(desc + 2, n + 9)
if rank == 0:
    assert m3 == 1907 + world - 1 and int(d3[0, 0]) == 11 + world - 1
    assert int(d2[0, 0]) == 10 + world - 1 and int(d2[5, 7]) == 200 + world - 1
else:
    assert m3 == 1909 + rank - 1 and int(d3[0, 0]) == 12 + rank - 1
    assert int(d2[0, 0]) == 11 + rank - 1
# a fourth call overwrites the buffer of the first: the rotation has three
d4, m4 = comm.exchange_boundary(desc + 3, n + 11)
assert (m4 == 1909 + world - 1) if rank == 0 else (m4 == 1911 + rank - 1)
# a new sequence (dvs_pipeline_reset's path): rank 0 has no predecessor again, the others see THIS call's blocks
comm.reset_sequence()
d5, m5 = comm.exchange_boundary(desc + 4, n + 13)
if rank == 0:
    assert d5 is None and m5 is None
else:
    assert m5 == 1913 + rank - 1 and int(d5[0, 0]) == 14 + rank - 1
# another capacity: the buffers are re-made and the sequence restarts
small = np.full((64, 32), 50 + rank, np.uint8)
d6, m6 = comm.exchange_boundary(small, 60 + rank)
assert (d6 is None) if rank == 0 else (m6 == 60 + rank - 1 and d6.shape == (64, 32) and int(d6[63, 31]) == 50 + rank - 1)
comm.close()
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_boundary_exchange_two_gloo_ranks(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    import socket
    with socket.socket() as so:     # a free port: a fixed one may still sit in TIME_WAIT from the previous test run
        so.bind(("127.0.0.1", 0))
        port = str(so.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", port, str(script), ROOT], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.count("ok") == 2


def test_boundary_exchange_three_gloo_ranks(tmp_path):
    """an odd world: rank 1 and rank 2 both read a neighbour's block of the same call, rank 0 wraps to rank 2's of the call before"""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = str(so.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
                          "--master-port", port, str(script), ROOT], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.count("ok") == 3


def test_single_rank_host_exchange_returns_the_previous_batch():
    """one rank, host transport (the callback has nothing to move): its own last frame of the batch before; a failing transport surfaces
    as an error of the C-ABI call, and the pipeline refuses a host-transport communicator"""
    from dvslam_amd import dist as dvdist, DvsError
    comm = dvdist.HostComm(0, 1, lambda buf, rank, nbytes: None)
    desc = (np.arange(64 * 32) % 251).astype(np.uint8).reshape(64, 32)
    d, n = comm.exchange_boundary(desc, 17)
    assert d is None and n is None                                     # first batch: no predecessor
    d, n = comm.exchange_boundary(desc + 1, 18)
    assert n == 17 and (d == desc).all()                               # one rank: its own last frame of the batch before
    comm.close()

    def broken(buf, rank, nbytes):
        raise RuntimeError("link down")
    bad = dvdist.HostComm(0, 1, broken)
    with pytest.raises(DvsError):
        bad.exchange_boundary(desc, 1)
    assert isinstance(bad.error, RuntimeError)
    bad.close()


def test_level_shards_cover_every_level_once():
    """SURVEY.md §8e, small batches: levels -> ranks balanced by pixel count; every level owned by exactly one rank"""
    from dvslam_amd import dist as dvdist
    px = [1280 * 720, 1067 * 600, 889 * 500, 741 * 417, 617 * 347, 514 * 289, 429 * 241, 357 * 201]
    for world in (1, 2, 3, 4, 8):
        masks = dvdist.level_shards(px, world)
        assert len(masks) == world
        union = 0
        for m in masks:
            assert union & m == 0
            union |= m
        assert union == 0xFF
        load = [sum(px[l] for l in range(8) if m >> l & 1) for m in masks]
        assert max(load) <= max(px[0], 1.34 * sum(px) / world)      # level 0 alone is 32 % of the work: the floor for >= 4 ranks
    assert dvdist.level_shards(px, 8) == [1 << l for l in range(8)]


def test_bench_self_launch_spawns_the_ranks():
    """`python bench.py --gpus 2` outside a launcher starts its own two ranks as a child process (before torch / the GPU is
    touched); --dry-launch makes the ranks report themselves instead of running GPU work"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"], capture_output=True, text=True,
                         env=env, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    ranks = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert sorted(r["rank"] for r in ranks) == [0, 1] and all(r["world"] == 2 and r["dry_launch"] for r in ranks)
    assert len({r["pid"] for r in ranks}) == 2
    # what the N > 1 self-check looks at: rank 1 reads rank 0's block of the same call, rank 0 wraps to the LAST rank's of the call before
    # (nothing at the start of a sequence); and the keys of the `rccl` object of the result line
    by = {r["rank"]: r for r in ranks}
    assert by[1]["boundary_predecessor"] == {"batch5": [0, 5], "batch0": [0, 0]}
    assert by[0]["boundary_predecessor"] == {"batch5": [1, 4], "batch0": None}
    assert {"nranks", "version", "transport", "exchange", "boundary_check", "allgather_bytes", "allgather_us"} <= set(by[0]["rccl_keys"])
    # under a launcher (the driver's form) the same file must NOT spawn again: it reads RANK / WORLD_SIZE from the environment
    env2 = dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29655")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dry-launch"], capture_output=True, text=True,
                         env=env2, timeout=120)
    assert out.returncode == 0 and json.loads(out.stdout.strip().splitlines()[-1])["world"] == 1


def test_bench_rccl_report_shape():
    """bench.py's `rccl` object: one verdict per rank, a mismatch named by rank, never an exception"""
    sys.path.insert(0, ROOT)
    import bench
    ok = [{"rank": r, "ok": True, "batch": 23, "queries": 2000, "train": 1990, "predecessor": {"rank": (r - 1) % 8, "batch": 23 - (r == 0)}} for r in range(8)]
    rep = bench.rccl_report(8, 22304, "RCCL", ok, 64832, 41.5)
    assert rep["nranks"] == 8 and rep["allgather_bytes"] == 8 * 64832 and rep["allgather_us"] == 41.5
    assert rep["boundary_check"]["ranks_checked"] == 8 and rep["boundary_check"]["result"] == "identical to the oracle on every rank"
    bad = [dict(v) for v in ok]; bad[3]["ok"] = False; bad[0] = {"rank": 0, "ok": False, "error": "boom"}
    assert bench.rccl_report(8, 22304, "RCCL", bad, 64832, None)["boundary_check"]["result"] == "MISMATCH on rank(s) 0, 3"
    assert [bench.boundary_predecessor(r, 8, 0) for r in (0, 1, 7)] == [None, (0, 0), (6, 0)]
    assert bench.boundary_predecessor(0, 8, 9) == (7, 8) and bench.frame_seed(9, 6, 7) == 1234 + 101 * 3 + 49
    json.dumps(rep)


def test_comm_c_abi_refuses_without_gpu(hiplib):
    """the RCCL exchange behind the C-ABI: symbols exist, the block is {cap x 32 descriptor bytes, int32 count} padded to 64 bytes, and
    without a device dvs_comm_create says DVS_ERR_NO_DEVICE instead of falling back to anything"""
    import ctypes as C
    from dvslam_amd import device_count
    for cap in (1, 500, 2024, 3024):
        assert hiplib.dvs_boundary_block_bytes(cap) == (cap * 32 + 4 + 63) // 64 * 64
    if device_count() == 0:
        h = C.c_void_p()
        ident = (C.c_uint8 * 128)()
        assert hiplib.dvs_comm_create(0, 0, 1, ident, C.byref(h)) == -5


@pytest.mark.gpu
def test_comm_exchange_single_rank_on_gpu(gpu, hiplib):
    """world = 1 RCCL communicator through the C-ABI: the exchange packs {descriptors, n}, all-gathers in place and hands back
    the predecessor's block — with one rank its own last frame of the batch before; the gather buffers are used in turn"""
    import ctypes as C
    from dvslam_amd import _lib
    from dvslam_amd import dist as dvdist
    comm = dvdist.Comm(0, 0, 1, lambda ident: ident)
    assert comm.rccl_version > 0
    cap = 2024
    rng = np.random.default_rng(5)
    st = _lib.stream_create(0)
    seen = []
    last = None
    for rnd in range(4):
        desc = rng.integers(0, 256, size=(cap, 32), dtype=np.uint8)
        n = np.array([1900 + rnd], np.int32)
        d_desc = _lib.DeviceBuffer(desc.nbytes).upload(desc); d_n = _lib.DeviceBuffer(4).upload(n)
        pd, pn = comm.exchange_boundary(st, d_desc.ptr, d_n.ptr, cap)
        _lib.stream_synchronize(st)   # the exchange is asynchronous on `st`, a non-blocking stream
        if rnd == 0:
            assert pd == 0 and pn == 0, "first batch: no predecessor"
        else:   # one rank: the predecessor of its first frame is its own last frame of the batch before
            got = np.empty((cap, 32), np.uint8); gn = np.empty(1, np.int32)
            _lib.check(hiplib.dvs_memcpy_d2h(0, got.ctypes.data, pd, got.nbytes)); _lib.check(hiplib.dvs_memcpy_d2h(0, gn.ctypes.data, pn, 4))
            assert (got == last[0]).all() and gn[0] == last[1]
        last = (desc, 1900 + rnd)
        seen.append(pd)
    assert seen[1] != seen[2] and seen[1] != 0, "gather buffers used in turn"
    comm.close()
    _lib.stream_destroy(st)
