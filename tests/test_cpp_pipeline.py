"""The C++ host of the benchmarked streaming step: tests/cpp/pipeline_stream.cpp drives
include/dvslam/streaming_pipeline.hpp -> dvs_pipeline_* (csrc/pipeline.hip) with plain g++, no Python in the loop.

* CPU: the program compiles -Wall -Werror against the C-ABI and refuses to run without a GPU (exit code 3).
* GPU: 5 pipelined steps of 64 x 1280x720 / 2000 kp from the C++ host — every resident frame and every match job equal the oracle
  (bit-exact), i.e. the C++ host's bytes are what tests/test_gpu_pipeline.py's Python caller of the same C-ABI produces;
  and configs[3] from C++: 8 logical ranks x 8 frames (dvs_comm_create_loopback, dvs_pipeline_attach_comm -> dvs_exchange_boundary)
  equal the single-rank 64-frame sequence."""
import os
import subprocess
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest
from dvslam_amd import synth
from dvslam_amd._lib import KP_DTYPE

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "dynamic-visual-slam_amd", "lib")
ROWS, COLS, NF = 720, 1280, 2000


def _build(tmpdir):
    exe = os.path.join(str(tmpdir), "pipeline_stream")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "pipeline_stream.cpp"), "-o", exe, "-L" + LIBDIR, "-ldvslam_hip", "-Wl,-rpath," + LIBDIR,
                           "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lpthread"])
    return exe


def test_cpp_pipeline_host_compiles_and_refuses_without_gpu(tmp_path, hiplib):
    from dvslam_amd import device_count
    exe = _build(tmp_path)
    if device_count() == 0:
        assert subprocess.call([exe]) == 3


def _read_out(path):
    """{(rank, step): (n, kps, desc, idx, dist)} as written by pipeline_stream.cpp"""
    raw = np.fromfile(path, np.uint8)
    world, B, cap, first, nw = raw[:20].view(np.int32)
    off = 20
    out = {}

    def take(nbytes):
        nonlocal off
        v = raw[off:off + nbytes]
        off += nbytes
        return v
    for r in range(world):
        for i in range(first, first + nw):
            n = take(B * 4).view(np.int32)
            kps = take(B * cap * 28).view(KP_DTYPE).reshape(B, cap)
            desc = take(B * cap * 32).reshape(B, cap, 32)
            idx = take(B * cap * 4).view(np.int32).reshape(B, cap)
            dist = take(B * cap * 4).view(np.int32).reshape(B, cap)
            out[(int(r), int(i))] = (n, kps, desc, idx, dist)
    assert off == raw.size
    return out, int(first), int(nw)


def _oracle_all(oracle, frames, threads=8):
    local = threading.local()

    def one(img):
        if not hasattr(local, "o"):
            local.o = oracle.OracleORB(NF, 1.2, 8, 20, 7)
        return local.o.extract(img)
    with ThreadPoolExecutor(threads) as ex:
        return list(ex.map(one, frames))


@pytest.mark.gpu
def test_cpp_host_runs_the_timed_configuration_against_oracle(tmp_path, gpu, oracle, hiplib):
    exe = _build(tmp_path)
    G, NB, STEPS, NSETS = 64, 2, 5, 4
    batches = [np.stack([synth.make_frame(i, COLS, ROWS, seed=1234 + 101 * (20 + g)) for i in range(G)]) for g in range(NB)]
    synth._CANVAS_CACHE.clear()
    fr = tmp_path / "frames.bin"
    np.concatenate(batches).tofile(fr)
    ref = [_oracle_all(oracle, b) for b in batches]
    # (a) one rank, 64 frames per step: bench.py's shape from a C++ host
    o1 = tmp_path / "one.bin"
    run = subprocess.run([exe, str(fr), str(G), str(ROWS), str(COLS), str(NF), str(NB), str(STEPS), str(NSETS), str(o1)], capture_output=True, text=True)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "pipeline_stream ok" in run.stdout
    one, first, nw = _read_out(o1)
    assert (first, nw) == (STEPS - NSETS, NSETS)
    jobs = 0
    for i in range(first, STEPS):
        n, k, d, idx, dist = one[(0, i)]
        for f in range(G):
            n2, k2, d2 = ref[i % NB][f]
            assert int(n[f]) == n2 and k[f, :n2].tobytes() == k2.tobytes() and (d[f, :n2] == d2).all(), (i, f)
            t = ref[i % NB][f - 1] if f else ref[(i - 1) % NB][G - 1]
            i2, dd2 = oracle.match(d2, t[2])
            assert (idx[f, :n2] == i2).all() and (dist[f, :n2] == dd2).all(), f"match job {f} of batch {i}"
            jobs += 1
    assert jobs == NSETS * G
    # (b) configs[3] from C++: 8 logical ranks x 8 frames through the loopback communicator = the single-rank sequence
    W, B = 8, 8
    o8 = tmp_path / "eight.bin"
    run = subprocess.run([exe, str(fr), str(B), str(ROWS), str(COLS), str(NF), str(NB), str(STEPS), str(NSETS), str(o8), str(W)], capture_output=True, text=True)
    assert run.returncode == 0, run.stdout + run.stderr
    eight, first8, nw8 = _read_out(o8)
    assert (first8, nw8) == (first, nw)
    for i in range(first, STEPS):
        n1, k1, d1, idx1, dist1 = one[(0, i)]
        for r in range(W):
            n, k, d, idx, dist = eight[(r, i)]
            for f in range(B):
                gf = r * B + f
                m = int(n[f])
                assert m == int(n1[gf]) and k[f, :m].tobytes() == k1[gf, :m].tobytes() and (d[f, :m] == d1[gf, :m]).all(), (i, r, f)
                assert (idx[f, :m] == idx1[gf, :m]).all() and (dist[f, :m] == dist1[gf, :m]).all(), f"match: batch {i} rank {r} frame {f}"
