#!/usr/bin/env python3
"""bench.py — frames/s of ORB extract + Hamming match at 1280x720 / 2000 keypoints (BASELINE.json
configs[1]) on N MI355X, with the roofline of the dominant kernel and the CPU oracle timed beside it.

A step = one pass of the hot path over one batch of B synthetic frames already resident in HBM:
pyramid -> FAST cells -> quad-tree -> blur -> orientation + rBRIEF for B frames, then B brute-force
match jobs (frame t vs t-1).  The input streams: the steps rotate over several resident batches of distinct
frames (more level-0 + pyramid bytes than the 256 MB Infinity Cache holds), and each step announces its real successor.
N > 1: one process per GPU, frames sharded contiguously over ranks, one RCCL all-gather of the boundary
descriptor block per step through the C-ABI (dvs_exchange_boundary).  `python bench.py --gpus N` without a
launcher starts its own N ranks (python -m torch.distributed.run) as a child process before anything touches the GPU.
Prints ONE JSON line on rank 0."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "dynamic-visual-slam_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

# algorithmic HBM bytes per 1280x720 frame, per stage (SURVEY.md §8d / BASELINE.md §3)
STAGE_BYTES = {
    "pyramid": 2_781_331 + 1_931_488,   # reads L0-L6, writes L1-L7
    "fast": 2_853_088,                  # reads L0-L7 once
    "octree": 0,                        # candidate lists only (cache resident, excluded from the headline)
    "blur": 2_853_088 + 2_853_088,      # reads + writes L0-L7
    "describe": 2000 * (28 + 32),       # keypoint + descriptor records
}
READ_BYTES_PER_FRAME = 8_487_507        # headline "HBM-read roofline" numerator
HBM_PEAK = 8.0e12


def cpu_baseline(frames, nfeatures, budget_s=12.0):
    """oracle (CPU restatement of the reference path) timed on the host, single thread, bounded sample"""
    import oracle_bindings as ob
    o = ob.OracleORB(nfeatures, 1.2, 8, 20, 7)
    L = ob.lib()
    n, k, d = o.extract(frames[0])          # warm-up, also gives the first "previous" descriptors
    prev = d
    t0 = time.perf_counter()
    done = 0
    for f in frames[1:]:
        n, k, d = o.extract(f)
        idx = np.zeros(len(d), np.int32); dist = np.zeros(len(d), np.int32)
        L.orc_match_hamming256(d.ctypes.data, len(d), prev.ctypes.data, len(prev), idx.ctypes.data, dist.ctypes.data)
        prev = d
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "frames/s", "cores": 1, "kind": "port", "flags": ob.FLAGS, "cpu": ob.cpu_model(),
            "sample": f"{done} frames 1280x720 extract(2000kp)+match vs previous frame, oracle/ (scalar C++ restatement of the "
                      "reference path: NOT OpenCV's SIMD kernels), 1 thread"}


def cpu_baseline_all_cores(frames, nfeatures, threads, budget_s=8.0):
    """the same oracle with frame-level parallelism (one extractor instance per thread): every thread walks
    the sample sequence extract + match-vs-previous on its own; ctypes releases the GIL inside the C++ calls"""
    import threading
    import oracle_bindings as ob
    L = ob.lib()
    counts = [0] * threads
    stop = time.perf_counter() + budget_s

    def worker(w):
        o = ob.OracleORB(nfeatures, 1.2, 8, 20, 7)
        n, k, prev = o.extract(frames[w % len(frames)])
        i = w
        while time.perf_counter() < stop:
            i += 1
            n, k, d = o.extract(frames[i % len(frames)])
            idx = np.zeros(len(d), np.int32); dist = np.zeros(len(d), np.int32)
            L.orc_match_hamming256(d.ctypes.data, len(d), prev.ctypes.data, len(prev), idx.ctypes.data, dist.ctypes.data)
            prev = d
            counts[w] += 1

    t0 = time.perf_counter()
    ts = [threading.Thread(target=worker, args=(w,)) for w in range(threads)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    dt = time.perf_counter() - t0
    return {"value": sum(counts) / dt, "unit": "frames/s", "cores": threads, "kind": "port", "flags": ob.FLAGS, "cpu": ob.cpu_model(),
            "sample": f"{sum(counts)} frames 1280x720 extract(2000kp)+match, oracle/ (scalar C++ restatement), {threads} threads (one extractor each)"}


def _replicate_ba(P, W):
    """W independent copies of one window as ONE block-diagonal problem: a batch of windows per launch pair"""
    Q = dict(P)
    K, L = P["K"], P["L"]
    Q["K"], Q["L"] = K * W, L * W
    for k in ("q", "t", "X", "uv"):
        Q[k] = np.tile(P[k], (W, 1))
    Q["cam_idx"] = np.concatenate([P["cam_idx"] + w * K for w in range(W)]).astype(np.int32)
    Q["lm_idx"] = np.concatenate([P["lm_idx"] + w * L for w in range(W)]).astype(np.int32)
    Q["pose_fixed"] = np.tile(P["pose_fixed"], W); Q["lm_fixed"] = np.tile(P["lm_fixed"], W)
    return Q


def ba_bench(dvslam_amd, synth, device, iters=200, W=64):
    """second half of BASELINE.json's metric: BA residual-evaluations/s on the 10 KF x 2000 LM window (config 3).
    One evaluation = all 20 000 residual blocks with local Jacobians, Huber corrector, cost and the H_pp / H_ll / g
    reductions (k_ba_eval + k_ba_reduce), parameters and outputs resident in HBM.  A single window is launch/latency
    bound (~4 MB, ~6 MFLOP), so the throughput figure batches W independent windows per launch pair (SURVEY.md §8e:
    "independent windows shard trivially"); the single-window latency is reported beside it."""
    import oracle_bindings as ob
    P = synth.make_ba_problem(K=10, L=2000, seed=42)
    R = len(P["cam_idx"])
    g = dvslam_amd.BAProblem(P, device=device)
    g.evaluate_device(20); g.synchronize()
    t0 = time.perf_counter(); g.evaluate_device(iters); g.synchronize(); dt1 = (time.perf_counter() - t0) / iters
    c1 = g.evaluate()[0]
    gb = dvslam_amd.BAProblem(_replicate_ba(P, W), device=device)
    cW = gb.evaluate()[0]
    assert abs(cW - W * c1) <= 1e-9 * abs(W * c1), "batched evaluation must equal W x the single-window cost"
    gb.evaluate_device(10); gb.synchronize()
    t0 = time.perf_counter(); gb.evaluate_device(iters // 2); gb.synchronize(); dtW = (time.perf_counter() - t0) / (iters // 2)
    bytes_eval = 528_560 + R * 160
    out = {"metric": "BA residual-eval/sec 10KF x 2000LM", "evals_per_s": round(W / dtW, 1), "windows_per_launch": W,
           "residual_blocks_per_s": round(W * R / dtW, 1), "us_per_eval_batched": round(1e6 * dtW / W, 3),
           "single_window_evals_per_s": round(1 / dt1, 1), "single_window_us_per_eval": round(1e6 * dt1, 2), "residual_blocks": R,
           "dtype": "f64", "algorithmic_bytes_per_eval": bytes_eval, "achieved_GBps": round(bytes_eval * W / dtW / 1e9, 2),
           "roofline_frac_hbm": round(bytes_eval * W / dtW / HBM_PEAK, 5),
           # the BASELINE config as stated is ONE window: launch-latency bound; the batched figure is W independent windows
           "roofline": {"single_window": {"bound": "hbm", "achieved": round(bytes_eval / dt1 / 1e9, 2), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                                          "frac": round(bytes_eval / dt1 / HBM_PEAK, 5), "us_per_eval": round(1e6 * dt1, 2)},
                        "batched": {"bound": "hbm", "achieved": round(bytes_eval * W / dtW / 1e9, 2), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                                    "frac": round(bytes_eval * W / dtW / HBM_PEAK, 5), "windows_per_launch": W}}}
    t0 = time.perf_counter(); s = g.solve(20); t_host_schur = time.perf_counter() - t0
    gd = dvslam_amd.BAProblem(P, device=device)
    gd.solve_device(20)                      # warm-up: a whole solve (kernels loaded, the runtime's launch resources grown)
    t_solves = []
    for _ in range(5):                       # median of five solves, each on a fresh problem (workspace allocation outside the timing)
        gd = dvslam_amd.BAProblem(P, device=device)
        gd.solve_device(0)
        t0 = time.perf_counter(); sd = gd.solve_device(20); t_solves.append(time.perf_counter() - t0)
    t_device = sorted(t_solves)[len(t_solves) // 2]
    out["lm_solve"] = {"iterations": s.num_iterations, "successful_steps": s.num_successful_steps, "initial_cost": s.initial_cost,
                       "final_cost": s.final_cost, "ms_gpu_eval_host_schur": round(1e3 * t_host_schur, 3),
                       "device": {"iterations": sd.num_iterations, "successful_steps": sd.num_successful_steps,
                                  "final_cost": sd.final_cost, "ms": round(1e3 * t_device, 3)}}
    o = ob.OracleBA(P)
    o.evaluate()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 3.0:
        o.evaluate(); n += 1
    cpu_rate = n / (time.perf_counter() - t0)
    o.evaluate_mt(4, 2)
    t0 = time.perf_counter(); n4 = 0
    while time.perf_counter() - t0 < 3.0:
        o.evaluate_mt(4, 8); n4 += 8
    cpu_rate4 = n4 / (time.perf_counter() - t0)
    o = ob.OracleBA(P)
    t1 = time.perf_counter(); so = o.solve(20); out["lm_solve"]["ms_cpu_oracle"] = round(1e3 * (time.perf_counter() - t1), 3)
    out["lm_solve"]["cpu_oracle_final_cost"] = so.final_cost
    out["cpu_baseline"] = {"value": round(cpu_rate, 2), "unit": "evals/s", "cores": 1, "kind": "port", "flags": ob.FLAGS, "cpu": ob.cpu_model(),
                           "sample": f"{n} evaluations of the same window with the oracle (Jet<double,10> autodiff as Ceres does), 1 thread"}
    out["cpu_baseline_4_threads"] = {"value": round(cpu_rate4, 2), "unit": "evals/s", "cores": 4, "kind": "port", "flags": ob.FLAGS,
                                     "sample": f"{n4} evaluations, residual blocks split over 4 threads as ceres::Solver::Options::num_threads = 4 "
                                               "(bundle_adjustment.hpp:842) does"}
    return out


_RESULT_FD = None


def _stdout_to_stderr():
    """Everything a rank writes to file descriptor 1 from here on goes to stderr — RCCL prints a version banner ("RCCL version : ...")
    on stdout when a communicator comes up — and the ONE JSON line leaves through the saved descriptor (_emit)."""
    global _RESULT_FD
    if _RESULT_FD is None:
        sys.stdout.flush()
        _RESULT_FD = os.dup(1)
        os.dup2(2, 1)


def _emit(obj):
    sys.stdout.flush()
    line = (json.dumps(obj) + "\n").encode()
    os.write(_RESULT_FD if _RESULT_FD is not None else 1, line)


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def self_launch(args, argv):
    """`python bench.py --gpus N` outside a launcher: start the N ranks ourselves (one process per GPU) as a CHILD process —
    nothing in this process has touched the GPU or imported torch — forward its output and exit with its code."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


def make_batches(synth, torch, dev, B, NB, rank, rows, cols, distinct=True):
    """NB resident batches of B frames, every frame of every batch its own image: batch b = B consecutive frames of the synthetic
    sequence over a scene of its own (seed per (rank, batch)).  Returns the device tensor [NB, B, rows, cols] and the frame count."""
    d = torch.empty((NB, B, rows, cols), dtype=torch.uint8, device=dev)
    for b in range(NB):
        seed = 1234 + 101 * b + 7 * rank
        for i in range(B):
            d[b, i].copy_(torch.from_numpy(synth.make_frame(i, cols, rows, seed=seed)))
        synth._CANVAS_CACHE.clear()
    return d, NB * B


def level_sharded_bench(args, world, rank, local, dev):
    """SURVEY.md section 8e, small batches: the SAME `--batch` frames on every rank (the broadcast of level 0 is the caller's and
    outside the timed region, like the resident input of the frame-sharded mode), levels sharded over the ranks, one all-gather
    per step, merge on every rank; the sequence match is split contiguously over the ranks (pair t on rank t*world//B)."""
    import torch
    import torch.distributed as dist
    import dvslam_amd
    from dvslam_amd import synth
    from dvslam_amd import dist as dvdist
    rows, cols, B, NB = 720, 1280, args.batch, max(1, args.resident_batches)
    d_img, frames_distinct = make_batches(synth, torch, dev, B, NB, 0, rows, cols, True)     # rank 0's seed on every rank
    orb = dvslam_amd.ORBextractor(args.nfeatures, 1.2, 8, 20, 7, device=local, max_batch=B)
    ts = torch.cuda.ExternalStream(orb.get_stream(), device=dev)
    mat = dvslam_amd.BFMatcher(device=local, stream=ts.cuda_stream)
    cap = orb.capacity
    with torch.cuda.stream(ts):
        kps = torch.empty((B, cap, 28), dtype=torch.uint8, device=dev); desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
        n = torch.zeros(B, dtype=torch.int32, device=dev)
        idx = torch.empty((B, cap), dtype=torch.int32, device=dev); dst = torch.empty((B, cap), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    comm = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
        dist.barrier()

        def bcast_id(ident):
            box = [ident]
            dist.broadcast_object_list(box, src=0)
            return box[0]
        comm = dvdist.Comm(local, rank, world, bcast_id)
        print(f"[bench] rank {rank}: RCCL communicator of {world} ranks (version {comm.rccl_version})", file=sys.stderr, flush=True)
    lx = dvdist.LevelShardedExtractor(orb, comm, rank, world, rows, cols, B)
    p0, p1 = rank * B // world, (rank + 1) * B // world       # this rank's match pairs (t, t-1), t in [max(p0,1), p1)
    state = {"i": 0}

    def step():
        i = state["i"]; state["i"] += 1
        with torch.cuda.stream(ts):
            lx.extract(d_img[i % NB].data_ptr(), rows, cols, cols, rows * cols, kps.data_ptr(), desc.data_ptr(), cap, n.data_ptr())
        t0 = max(p0, 1)
        if p1 > t0:
            mat.match_sequence_device(desc[t0].data_ptr(), n[t0:].data_ptr(), cap, p1 - t0, desc[t0 - 1].data_ptr(), n[t0 - 1:].data_ptr(),
                                      idx[t0].data_ptr(), dst[t0].data_ptr())

    def barrier():
        ts.synchronize(); torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        _emit({
            "metric": "frames/sec ORB+match @1280x720x2000kp", "value": round(B * args.steps / elapsed, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "1280x720 gray frames, ORBextractor(2000,1.2,8,20,7) extract + BFMatcher(HAMMING) match vs previous frame, "
                                   "small batch", "frames_per_step_total": B, "frames_distinct": frames_distinct,
                       "parallelism": f"level-sharded x{world}: masks {[hex(m) for m in lx.masks]}, all_gather of {lx.block_bytes} B blocks, merge on "
                                      f"every rank", "keypoints_frame0": int(n[0].item())},
            "rccl": None if comm is None else {"nranks": world, "version": comm.rccl_version}})
    if comm is not None:
        torch.cuda.synchronize()
        comm.close()
    if dist.is_initialized():
        dist.destroy_process_group()


def oracle_check(synth, pipe, batch_of_step, rows, cols, nfeatures, sample=(0, 1, -1)):
    """after the timed region: frames of the LAST extracted batch and match jobs of the batch before it — results of the very schedule
    that was timed — against the CPU oracle (the checker: tests/oracle_bindings.py, never part of the timed path)"""
    import oracle_bindings as ob
    B, i_last = pipe.B, pipe.i - 1
    n, kps, desc = pipe.outputs(i_last)
    o = ob.OracleORB(nfeatures, 1.2, 8, 20, 7)
    bad = []
    frames = sorted({f % B for f in sample})
    for f in frames:
        n2, k2, d2 = o.extract(batch_of_step(i_last, f))
        if not (int(n[f]) == n2 and kps[f, :n2].tobytes() == k2.tobytes() and (desc[f, :n2] == d2).all()):
            bad.append(f"frame {f} of batch {i_last}")
    jobs = 0
    if i_last >= 1:
        j = i_last - 1
        nj, _, dj = pipe.outputs(j)
        idx, dist = pipe.matches(j)
        for f in [f for f in frames if f >= 1]:
            i2, d2 = ob.match(dj[f, :nj[f]], dj[f - 1, :nj[f - 1]])
            jobs += 1
            if not ((idx[f, :nj[f]] == i2).all() and (dist[f, :nj[f]] == d2).all()):
                bad.append(f"match job {f} of batch {j}")
    return {"frames": len(frames), "match_jobs": jobs, "result": "identical to the oracle" if not bad else "MISMATCH: " + ", ".join(bad)}


def frame_seed(batch_index, NB, rank):
    """seed of the scene that rank `rank` extracts in step `batch_index` (make_batches: one scene per (rank, resident batch))"""
    return 1234 + 101 * (batch_index % NB) + 7 * rank


def boundary_predecessor(rank, world, j):
    """(rank, batch) whose LAST frame precedes this rank's frame 0 of batch j in the global frame order — what dvs_exchange_boundary must
    have delivered: the previous rank's block of the same call, or for rank 0 the last rank's block of the call before (None: the
    sequence starts there).  The reference's analogue is prev_descriptors_ of frontend.cpp:1096-1132."""
    if rank > 0:
        return rank - 1, j
    return (world - 1, j - 1) if j >= 1 else None


def boundary_verdict(synth, pipe, rank, world, NB, rows, cols, nfeatures, j):
    """N > 1 self-check, run on EVERY rank after the timed region: match job 0 of batch j — the only job that depends on the all-gather —
    against the oracle's match of this rank's own frame-0 descriptors with the ORACLE's extraction of the predecessor frame, which every
    rank can synthesise from the seeds.  Never raises: a failure is a verdict."""
    try:
        import oracle_bindings as ob
        nj, _, dj = pipe.outputs(j)
        idx, dst = pipe.matches(j)
        nq = int(nj[0])
        pred = boundary_predecessor(rank, world, j)
        if pred is None:
            ok = bool((idx[0, :nq] == -1).all())
            return {"rank": rank, "ok": ok, "batch": j, "queries": nq, "train": 0, "predecessor": None}
        pr, pb = pred
        frame = synth.make_frame(pipe.B - 1, cols, rows, seed=frame_seed(pb, NB, pr))
        n2, _, d2 = ob.OracleORB(nfeatures, 1.2, 8, 20, 7).extract(frame)
        i2, dd2 = ob.match(dj[0, :nq], d2)
        ok = bool((idx[0, :nq] == i2).all() and (dst[0, :nq] == dd2).all())
        return {"rank": rank, "ok": ok, "batch": j, "queries": nq, "train": int(n2), "predecessor": {"rank": pr, "batch": pb}}
    except Exception as e:   # noqa: BLE001
        return {"rank": rank, "ok": False, "error": repr(e)}


def rccl_report(world, version, transport, verdicts, block_bytes, allgather_us):
    """the `rccl` object of the result line: every rank's boundary verdict, gathered on rank 0"""
    bad = [v for v in verdicts if not v.get("ok")]
    return {"nranks": world, "version": version, "transport": transport,
            "exchange": "dvs_exchange_boundary: one in-place all-gather of the ranks' last-frame blocks per global batch, behind the C-ABI",
            "boundary_check": {"ranks_checked": len(verdicts), "job": "frame 0 of the last matched batch against the predecessor the exchange delivered, "
                                                                      "vs the oracle on the oracle's extraction of that predecessor frame",
                               "result": "identical to the oracle on every rank" if not bad else
                                         "MISMATCH on rank(s) " + ", ".join(str(v.get("rank")) for v in bad),
                               "per_rank": verdicts},
            "allgather_bytes": int(world * block_bytes), "allgather_us": allgather_us}


def time_exchange(pipe, comm, stream_sync, reps=20):
    """wall time of one dvs_exchange_boundary (pack + all-gather, nothing else on the GPU), after the timed region; collective: every rank
    makes the same calls"""
    from dvslam_amd import _lib
    s = (pipe.i - 1) % pipe.nsets
    pd, pn = pipe._last(s)
    st = _lib.stream_create(pipe.device)   # (the pipeline is drained: any stream will do, and its own are the library's business)
    stream_sync()
    try:
        for _ in range(3):
            comm.exchange_boundary(st, pd, pn, pipe.cap)
        _lib.stream_synchronize(st)
        t0 = time.perf_counter()
        for _ in range(reps):
            comm.exchange_boundary(st, pd, pn, pipe.cap)
        _lib.stream_synchronize(st)
        return round((time.perf_counter() - t0) / reps * 1e6, 1)
    finally:
        _lib.stream_synchronize(st)
        _lib.stream_destroy(st)


def loopback_bench(args):
    """`--loopback N`: the frame-sharded step of N ranks on ONE GPU — N pipelines, each driven by its own host thread, joined by the loopback
    communicator (same dvs_exchange_boundary, a host rendezvous + device-to-device pulls in RCCL's place).  A rehearsal of the N > 1
    code path (sharding, exchange, the boundary self-check and its report), not a measurement: the line says DIAGNOSTIC."""
    import threading
    _stdout_to_stderr()
    import torch
    from dvslam_amd import synth
    from dvslam_amd import dist as dvdist
    from dvslam_amd.pipeline import StreamingPipeline
    assert torch.cuda.is_available(), "bench.py needs the MI355X (no CPU fallback)"
    N, rows, cols = args.loopback, 720, 1280
    assert args.global_batch % N == 0
    B = args.global_batch // N
    NB = min(2, max(1, args.resident_batches))
    dev = torch.device("cuda", 0)
    imgs = []
    for r in range(N):
        d, _ = make_batches(synth, torch, dev, B, NB, r, rows, cols, True)
        imgs.append(d)
    pipes = [StreamingPipeline(B, rows, cols, args.nfeatures, device=0, nsets=args.nsets, pipelined=args.pipelined, lanes=args.lanes,
                               quadtree_async=args.quadtree_async) for _ in range(N)]
    comms = dvdist.Comm.loopback(0, N)
    for p_, c_ in zip(pipes, comms):
        p_.attach_comm(c_)
    torch.cuda.synchronize()
    gate = threading.Barrier(N)
    elapsed, ag_us, verdicts, errors = [0.0] * N, [None] * N, [None] * N, []

    def run(r):
        try:
            pipe, img = pipes[r], [imgs[r][k].data_ptr() for k in range(NB)]
            for _ in range(args.warmup):
                pipe.step(img[pipe.i % NB], img[(pipe.i + 1) % NB])
            pipe.synchronize(); gate.wait()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                pipe.step(img[pipe.i % NB], img[(pipe.i + 1) % NB])
            pipe.synchronize(); gate.wait()
            elapsed[r] = time.perf_counter() - t0
            ag_us[r] = time_exchange(pipe, comms[r], pipe.synchronize, reps=5)
            jb = pipe.i - 2 if args.pipelined else pipe.i - 1
            verdicts[r] = boundary_verdict(synth, pipe, r, N, NB, rows, cols, args.nfeatures, jb)
        except Exception as e:   # noqa: BLE001
            errors.append((r, repr(e)))
            gate.abort()
    th = [threading.Thread(target=run, args=(r,)) for r in range(N)]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join()
    el = max(elapsed)
    cap = pipes[0].cap
    rep_ = rccl_report(N, 0, "loopback group (dvs_comm_create_loopback): host rendezvous + device-to-device pulls in RCCL's place",
                       [v if v is not None else {"rank": r, "ok": False, "error": "rank did not finish"} for r, v in enumerate(verdicts)],
                       pipes[0].L.dvs_boundary_block_bytes(cap), max([u for u in ag_us if u is not None], default=None))
    out = {"metric": "DIAGNOSTIC (loopback rehearsal: N logical ranks on ONE GPU) frames/sec ORB+match @1280x720x2000kp",
           "diagnostic": "not a scaling result: the ranks share one GPU; what it rehearses is the N > 1 code path and its self-check",
           "value": round(N * B * args.steps / el, 2) if el > 0 else 0.0, "unit": "frames/s", "n_gpus": 1, "logical_ranks": N, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(1e3 * el / max(args.steps, 1), 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u8",
           "data": "synthetic", "config": {"workload": f"{args.global_batch} frame pairs per step over {N} logical ranks ({B} each), 1280x720, 2000 kp",
                                           "frames_per_rank_per_step": B}, "rccl": rep_, "errors": errors}
    for c_ in comms:
        c_.close()
    _emit(out)
    if "MISMATCH" in rep_["boundary_check"]["result"] or errors:
        print(f"[bench] WARNING: loopback rehearsal: {rep_['boundary_check']['result']} {errors}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=64, help="frames per GPU per step (weak scaling: fixed per GPU)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="strong: BASELINE configs[3] as written — `--global-batch` frame pairs IN TOTAL per step, split contiguously over the ranks "
                         "(8 GPUs: 8 frames per GPU per step); weak (default): `--batch` frames per GPU per step")
    ap.add_argument("--global-batch", type=int, default=64, help="--scaling strong: frames per step over all ranks")
    ap.add_argument("--quadtree-async", type=int, default=0, choices=(-1, 0, 1),
                    help="dvs_pipeline_params::quadtree_async: the four-stream form of the two-stream pipeline on (1) / off (-1) / by batch size (0)")
    ap.add_argument("--lanes", type=int, default=0, help="dvs_pipeline_params::lanes: 0 = by batch size, 1 = two-stream software pipeline, 2..4 = lane schedule")
    ap.add_argument("--nsets", type=int, default=0, help="output sets of the pipeline in rotation (0 = the library's choice: 4, or two per lane)")
    ap.add_argument("--nfeatures", type=int, default=2000)
    ap.add_argument("--serial-match", dest="pipelined", action="store_false",
                    help="match batch i in step i behind its own extraction on one stream.  Default: software-pipelined (dvslam_amd/pipeline.py) "
                         "— step i extracts batch i and matches batch i - 1 on a second stream that the extractor releases behind FAST, the "
                         "descriptor stage of batch i runs beside FAST of batch i + 1; every step still runs one extraction and one match of "
                         "64 frames")
    ap.add_argument("--resident-batches", type=int, default=6,
                    help="distinct resident input batches the steps rotate over (6 x 64 x 0.92 MB of level 0 + 6 x 180 MB of pyramids and "
                         "blurred levels per pass: far beyond the 256 MB Infinity Cache)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--shard", choices=("frames", "levels"), default="frames",
                    help="levels: SURVEY.md section 8e's small-batch mode (use with --batch < 8): every rank holds the same frames, extracts its "
                         "own pyramid levels, one all-gather of level-slotted blocks, on-device merge; total work fixed (strong scaling)")
    ap.add_argument("--loopback", type=int, default=0, help="rehearsal of the N > 1 path on ONE GPU: N logical ranks of this process "
                    "(dvs_comm_create_loopback, a host thread each), --global-batch frames split over them; prints a DIAGNOSTIC line with the same `rccl` object")
    ap.add_argument("--dry-launch", action="store_true", help="only start the ranks and report them (no GPU work): launcher self-test")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if args.dry_launch:
        # (also what the N > 1 self-check would look at on this rank, and the keys of the `rccl` object it fills: no GPU work)
        _emit({"dry_launch": True, "rank": rank, "world": world, "local_rank": local, "pid": os.getpid(),   # (one write: the ranks share a pipe)
               "boundary_predecessor": {"batch5": boundary_predecessor(rank, world, 5), "batch0": boundary_predecessor(rank, world, 0)},
               "rccl_keys": sorted(rccl_report(world, 0, "dry launch", [{"rank": rank, "ok": True}], 64, None))})
        return
    if args.loopback >= 2:
        return loopback_bench(args)

    _stdout_to_stderr()
    import torch
    import torch.distributed as dist
    import dvslam_amd
    from dvslam_amd import synth, _lib
    from dvslam_amd import dist as dvdist
    from dvslam_amd.pipeline import StreamingPipeline

    assert torch.cuda.is_available(), "bench.py needs the MI355X (no CPU fallback)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if args.shard == "levels":
        return level_sharded_bench(args, world, rank, local, dev)
    rows, cols, B = 720, 1280, args.batch
    if args.scaling == "strong":
        assert args.global_batch % world == 0, f"--global-batch {args.global_batch} does not split over {world} ranks"
        B = args.global_batch // world
    NB = max(1, args.resident_batches)
    # synthetic input, resident in HBM before the timed region: this rank's shard of NB global batches
    d_img, frames_distinct = make_batches(synth, torch, dev, B, NB, rank, rows, cols, True)
    img = [d_img[k].data_ptr() for k in range(NB)]

    # One pipeline per GPU (dvslam_amd/pipeline.py: the step that tests/test_gpu_pipeline.py checks against the oracle).  All its streams
    # are created by the library, back to back, BEFORE RCCL comes up.
    pipe = StreamingPipeline(B, rows, cols, args.nfeatures, device=local, nsets=args.nsets, pipelined=args.pipelined,
                             lanes=args.lanes, quadtree_async=args.quadtree_async)
    cap = pipe.cap
    torch.cuda.synchronize()
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ   # under torchrun (also with one rank): exercise RCCL
    comm = None
    rccl = None
    if world > 1 or launched:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
        dist.barrier()

        def bcast_id(ident):   # the out-of-band hand-over of the RCCL unique id: torch.distributed's store
            box = [ident]
            dist.broadcast_object_list(box, src=0)
            return box[0]
        comm = dvdist.Comm(local, rank, world, bcast_id)
        pipe.attach_comm(comm)
        rccl = {"nranks": world, "version": comm.rccl_version, "exchange": "dvs_exchange_boundary: ncclAllGather behind the C-ABI"}   # (completed after the run)
        print(f"[bench] rank {rank}: RCCL communicator of {world} ranks (version {comm.rccl_version})", file=sys.stderr, flush=True)
    torch.cuda.synchronize()

    diag_no_match = os.environ.get("BENCH_DIAG_NO_MATCH") == "1"   # diagnostics only: the step without its match (the line says so)

    def step():
        i = pipe.i
        if diag_no_match:
            pipe.step(img[i % NB], img[(i + 1) % NB], match=False)
        else:
            pipe.step(img[i % NB], img[(i + 1) % NB])

    def barrier():
        pipe.synchronize()
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    # Per-kernel durations FIRST (they are part of this benchmark's report and leave the GPU at its working clocks: the timed region
    # of a short run — the driver's `--steps 20 --warmup 5` — otherwise starts on a GPU that idled through the seconds of host-side frame
    # synthesis above and runs its first ~25 steps 10-15 % slower, profiles/r03_steps_warmup_sweep.json).  K steps each, hipEvents
    # around every stage launch on the stream it runs on (the events cost ~10 us of stream time per stage, so they never share the
    # whole-job timing below):
    # (b) every kernel alone on the stream: the kernel's own duration, which the rooflines below are computed from
    K = args.steps
    overlap_default = os.environ.get("DVS_NO_OVERLAP") != "1"   # tools/collect_profiles.sh serialises the WHOLE run for its per-kernel passes
    lanes = pipe.lanes                                          # >= 2: the small-batch lane schedule; its extractors have one stream each
    pipe.stage_timing(True)                                     # (lane 0's extractor: every lanes-th step is timed)
    if lanes >= 2:
        for k in range(K * lanes):
            pipe.step(img[k % NB], 0, match=False)
            pipe.synchronize()                                  # one step in flight: every kernel alone on the machine
    else:
        pipe.set_serialized(True)
        for k in range(K):
            pipe.step(img[k % NB], 0, match=False)
    pipe.synchronize()
    stage_ms, stage_calls = pipe.stage_times()
    if lanes < 2:
        pipe.set_serialized(not overlap_default)
    # (a) the pipelined schedule: what a kernel takes WHILE its neighbours share the machine (= rocprofv3's statistics of this command);
    #     last, so that the timed region follows work of its own intensity
    pipe.reset()
    for k in range(K * max(lanes, 1)):
        pipe.step(img[k % NB], img[(k + 1) % NB])
    pipe.synchronize()
    ov_ms, ov_calls = pipe.stage_times()
    pipe.stage_timing(False)
    pipe.reset()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    host_enqueue = time.perf_counter() - t0   # host time to enqueue all steps (no synchronisation inside)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # checks of what was just timed (outside the timed region)
    match_check = ocheck = None
    if comm is not None:
        # N > 1 — and one rank under the launcher, a real 1-rank RCCL: EVERY rank checks the one match job that depends on the all-gather
        # (frame 0 of a batch against the predecessor dvs_exchange_boundary delivered) against the oracle, rank 0 gathers the verdicts.
        # A mismatch prints loudly and never loses the result line.
        ag_us = None
        try:
            ag_us = time_exchange(pipe, comm, lambda: (pipe.synchronize(), torch.cuda.synchronize()))
        except Exception as e:   # noqa: BLE001
            print(f"[bench] rank {rank}: timing the exchange failed: {e!r}", file=sys.stderr, flush=True)
        jb = pipe.i - 2 if args.pipelined else pipe.i - 1
        verdict = (boundary_verdict(synth, pipe, rank, world, NB, rows, cols, args.nfeatures, jb) if jb >= 0 else
                   {"rank": rank, "ok": False, "error": "no matched batch to check"})
        verdicts = [verdict]
        try:
            box = [None] * world
            dist.all_gather_object(box, verdict)
            verdicts = box
        except Exception as e:   # noqa: BLE001
            verdicts = [dict(verdict, gather_error=repr(e))]
        rccl = rccl_report(world, comm.rccl_version, "RCCL ncclAllGather (dlopen'ed librccl.so.1)", verdicts, pipe.L.dvs_boundary_block_bytes(cap), ag_us)
        if rank == 0 and "MISMATCH" in rccl["boundary_check"]["result"]:
            print(f"[bench] WARNING: boundary check: {rccl['boundary_check']['result']}: {verdicts}", file=sys.stderr, flush=True)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle_bindings as ob
        ob.use_native()   # the CPU baseline below times a -O2 -march=native build made here; it must be selected before the first oracle call
    n_last = pipe.outputs(pipe.i - 1)[0] if pipe.i >= 1 else np.zeros(B, np.int32)
    matched = 0
    if rank == 0 and pipe.i >= 3:
        # (1) the pipelined match against the same job enqueued serially: guards the event dependencies
        j = pipe.i - 2 if args.pipelined else pipe.i - 1
        sj, sp = j % pipe.nsets, (j - 1) % pipe.nsets
        idx, dst = pipe.matches(j)
        nj = pipe.outputs(j)[0]
        if comm is None:
            i2 = _lib.DeviceBuffer(B * cap * 4, local); d2 = _lib.DeviceBuffer(B * cap * 4, local)
            pd, pn = pipe._last(sp)
            chk = dvslam_amd.BFMatcher(device=local)   # a matcher of its own: the pipeline is drained, its handles are the library's business
            chk.match_sequence_device(pipe.desc[sj].ptr, pipe.n[sj].ptr, cap, B, pd, pn, i2.ptr, d2.ptr)
            chk.synchronize()
            ri = i2.download(np.int32, B * cap).reshape(B, cap); rd = d2.download(np.int32, B * cap).reshape(B, cap)
            same = all((ri[f, :nj[f]] == idx[f, :nj[f]]).all() and (rd[f, :nj[f]] == dst[f, :nj[f]]).all() for f in range(B))
            match_check = "identical to the serial match of the same batch" if same else "MISMATCH"
            if not same:
                print("[bench] WARNING: the pipelined match differs from the same job enqueued serially", file=sys.stderr, flush=True)
        matched = int((dst[1, :nj[1]] < 50).sum()) if B > 1 else 0
        # (2) a sample of the timed schedule's own results against the CPU oracle (never allowed to lose the result line: ADVICE r3)
        if not args.no_cpu_baseline:
            try:
                seed_of = lambda i: 1234 + 101 * (i % NB) + 7 * rank
                ocheck = oracle_check(synth, pipe, lambda i, f: synth.make_frame(f, cols, rows, seed=seed_of(i)), rows, cols, args.nfeatures,
                                      sample=(0, 1, B // 2, B - 1))
                if "MISMATCH" in ocheck["result"]:
                    print(f"[bench] WARNING: {ocheck['result']}", file=sys.stderr, flush=True)
            except Exception as e:   # noqa: BLE001
                ocheck = {"error": repr(e)}

    if rank == 0:
        total_frames = world * B * args.steps
        fps = total_frames / elapsed
        dom = max(stage_ms, key=lambda k: stage_ms[k])
        dom_ms = stage_ms[dom] / max(stage_calls[dom], 1)
        # HBM bytes and VALU instructions per launch come from the committed PMC passes (profiles/pmc_traffic.json, scaled to this batch);
        # they are only quoted while the kernel sources are the ones the counters were collected on
        traffic = issue = None
        traffic_src = {"file": "profiles/pmc_traffic.json"}
        try:
            traffic_src["csrc_digest_now"] = _lib.kernel_source_digest()
            tj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            traffic_src["csrc_digest_at_collection"] = tj.get("csrc_digest")
            traffic_src["collected_at_commit"] = tj.get("commit")
            if tj.get("csrc_digest") == traffic_src["csrc_digest_now"]:
                traffic = int(tj["bytes_per_launch"][dom] * B / tj["batch"])
                insts = tj["valu_insts_per_launch"][dom] * B / tj["batch"]
                rate = insts / (dom_ms * 1e-3) / 1e9
                nominal = 256 * 4 * 2.4 / 2   # 1 024 SIMD-32 x one wave64 instruction per 2 cycles at 2.4 GHz (MI355X_MICROARCH.md)
                issue = {"kernel": dom, "valu_wave_insts_per_launch": int(insts), "achieved": round(rate, 1), "peak": nominal,
                         "unit": "G wave-instr/s", "frac": round(rate / nominal, 4),
                         "measured_class_rates_G_per_s": {"add/xor/max_i16 class": "780-950", "perm/pk16/dot/bcnt/min3/mad24 class": tj["valu_issue_peak_G_per_s"]},
                         "source": "SQ_INSTS_VALU from profiles/pmc_traffic.json; class rates: tools/ubench/valu_rate.hip, profiles/r02_valu_issue_rates.txt"}
            else:
                traffic_src["stale"] = True
                print("[bench] WARNING: csrc/ changed since profiles/pmc_traffic.json was collected: roofline.traffic = null "
                      "(re-run tools/collect_profiles.sh + tools/make_pmc_traffic.py)", file=sys.stderr, flush=True)
        except Exception as e:   # noqa: BLE001
            traffic_src["error"] = repr(e)
        achieved = STAGE_BYTES[dom] * B / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        sched = ("serial match behind its own extraction" if not args.pipelined else
                 f"{lanes} lanes: whole steps in flight on independent extractor / matcher pairs, one stream each" if lanes >= 2 else
                 "four streams: blur + FAST | next level chain | quad-tree | descriptors + match of batch i - 1" if pipe.quadtree_async else
                 "step i = extraction of batch i + match of batch i - 1 released behind FAST, descriptor stage of batch i beside FAST of batch i + 1")
        out = {
            "metric": "frames/sec ORB+match @1280x720x2000kp", "value": round(fps, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            # untimed steps ahead of the timed region: the two per-stage timing passes (they also bring the GPU to its working clocks) + warm-up
            "pre_timed_steps": 2 * K * max(lanes, 1) + args.warmup,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "1280x720 gray frames, ORBextractor(2000,1.2,8,20,7) extract + BFMatcher(HAMMING) match vs previous frame "
                                   + ("(BASELINE configs[1])" if args.scaling == "weak" else
                                      f"(BASELINE configs[3] as written: {args.global_batch} frame pairs per step IN TOTAL over {world} GPU(s))"),
                       "frames_per_gpu_per_step": B, "frames_per_step_total": B * world, "frames_distinct": frames_distinct,
                       "resident_batches": NB, "keypoints_frame1": int(n_last[min(1, B - 1)]), "matches_lt50_frame1": matched,
                       "parallelism": f"frame-sharded x{world}, boundary-descriptor all_gather, {sched}"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": round(achieved * 1e9 / HBM_PEAK, 5), "traffic": traffic, "traffic_source": traffic_src,
                         "ms_per_launch": round(dom_ms, 4), "algorithmic_bytes_per_launch": STAGE_BYTES[dom] * B},
            "valu_issue_roofline": issue,
            "hbm_read_roofline_frac": round(fps / world * READ_BYTES_PER_FRAME / HBM_PEAK, 5),
            "stage_ms_per_launch_isolated": {k: round(v / max(stage_calls[k], 1), 4) for k, v in stage_ms.items()},
            "stage_ms_per_launch_overlapped": {k: round(v / max(ov_calls[k], 1), 4) for k, v in ov_ms.items()},
            # wall time of the enqueue loop: the host's own cost only while it is ahead of the GPU — once the hardware queues are full
            # the runtime holds the caller back and this follows ms_per_step (tools/host_probe.sh: the HIP calls of one step)
            "host_enqueue_ms_per_step": round(host_enqueue / args.steps * 1e3, 4), "host": "C++ (dvs_pipeline_step: one C-ABI call per step)",
            "rccl": rccl, "match_check": match_check,
            "oracle_check": ocheck,
        }
        if diag_no_match:
            out["diagnostic"] = "BENCH_DIAG_NO_MATCH=1: the match was NOT enqueued in the timed region — not a result, a diagnostic"
            out["metric"] = "DIAGNOSTIC (extraction only) " + out["metric"]
        if world == 1 and not args.no_cpu_baseline:
            cb_frames = [synth.make_frame(t, cols, rows) for t in range(min(48, 64))]
            out["cpu_baseline"] = cpu_baseline(cb_frames, args.nfeatures)
            out["speedup_vs_cpu_1thread"] = round(fps / out["cpu_baseline"]["value"], 1)
            nthr = max(1, min(16, len(os.sched_getaffinity(0))))   # the GPU box's CPU share for one GPU
            out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(cb_frames, args.nfeatures, nthr)
            out["speedup_vs_cpu_all_cores"] = round(fps / out["cpu_baseline_all_cores"]["value"], 1)
            out["ba"] = ba_bench(dvslam_amd, synth, local)
        _emit(out)
    pipe.close()
    if comm is not None:
        torch.cuda.synchronize()
        comm.close()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
