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


def make_batches(synth, torch, dev, B, NB, rank, rows, cols, distinct):
    """NB resident batches of B frames.  distinct: every frame of every batch is its own image (batch b = 64 consecutive frames of
    the synthetic sequence over a scene of its own, seed per (rank, batch)); otherwise round 1's start-up-saving mode (16 frames
    tiled over ONE batch).  Returns the device tensor [NB, B, rows, cols] and the number of distinct frames."""
    if not distinct:
        uniq = min(B, 16)
        host = [synth.make_frame((rank * B + i) % 64, cols, rows) for i in range(uniq)]
        d = torch.empty((1, B, rows, cols), dtype=torch.uint8, device=dev)
        for i in range(B):
            d[0, i].copy_(torch.from_numpy(host[i % uniq]))
        return d, uniq
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
        print(json.dumps({
            "metric": "frames/sec ORB+match @1280x720x2000kp", "value": round(B * args.steps / elapsed, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "1280x720 gray frames, ORBextractor(2000,1.2,8,20,7) extract + BFMatcher(HAMMING) match vs previous frame, "
                                   "small batch", "frames_per_step_total": B, "frames_distinct": frames_distinct,
                       "parallelism": f"level-sharded x{world}: masks {[hex(m) for m in lx.masks]}, all_gather of {lx.block_bytes} B blocks, merge on "
                                      f"every rank", "keypoints_frame0": int(n[0].item())},
            "rccl": None if comm is None else {"nranks": world, "version": comm.rccl_version}}), flush=True)
    if comm is not None:
        torch.cuda.synchronize()
        comm.close()
    if dist.is_initialized():
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=64, help="frames per GPU per step")
    ap.add_argument("--nfeatures", type=int, default=2000)
    ap.add_argument("--match-stream", dest="match_stream", action="store_true",
                    help="give the match its own stream so that it can overlap the next batch's extraction (default: enqueued behind its "
                         "own batch's extraction on the same stream; measured faster, DESIGN.md section 5)")
    ap.add_argument("--serial-match", dest="match_late", action="store_false",
                    help="match batch i in step i behind its own extraction (round 1's schedule).  Default: software-pipelined — step i "
                         "extracts batch i and matches batch i - 1 on a second stream that the extractor releases behind FAST "
                         "(dvs_orb_set_after_fast_event), so the matrix-core match runs beside the quad-tree / blur phase; every step "
                         "still runs one extraction and one match of 64 frames")
    ap.add_argument("--defer", choices=("on", "off"), default="on",
                    help="deferred descriptor stage (dvs_orb_set_output_event + dvs_orb_set_defer_outputs): the next batch's FAST runs beside "
                         "this batch's descriptor gathers.  Measured +13 / +12 / +8.5 / +3.8 / +6.7 %% at 1 / 8 / 32 / 64 / 128 frames per step")
    ap.add_argument("--resident-batches", type=int, default=6,
                    help="distinct resident input batches the steps rotate over (6 x 64 x 0.92 MB of level 0 + 6 x 180 MB of pyramids and "
                         "blurred levels per pass: far beyond the 256 MB Infinity Cache)")
    ap.add_argument("--single-resident-batch", action="store_true", help="round 1's mode: 16 distinct frames tiled over one resident batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prefetch", dest="prefetch", action="store_false",
                    help="build every batch's pyramid inside its own step (default: the next batch's pyramid overlaps this batch's "
                         "FAST, dvs_orb_hint_next_batch_device)")
    ap.add_argument("--torch-exchange", action="store_true", help="exchange through torch.distributed (dist.py) instead of the C-ABI")
    ap.add_argument("--pipes", type=int, default=1, help="independent extractor/matcher pipelines per GPU; step i runs on pipeline i %% pipes "
                    "(experiment: lets one batch's FAST fill the issue slots the other batch's quad-tree / descriptor / match leave idle)")
    ap.add_argument("--shard", choices=("frames", "levels"), default="frames",
                    help="levels: SURVEY.md section 8e's small-batch mode (use with --batch < 8): every rank holds the same frames, extracts its "
                         "own pyramid levels, one all-gather of level-slotted blocks, on-device merge; total work fixed (strong scaling)")
    ap.add_argument("--dry-launch", action="store_true", help="only start the ranks and report them (no GPU work): launcher self-test")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if args.dry_launch:
        print(json.dumps({"dry_launch": True, "rank": rank, "world": world, "local_rank": local, "pid": os.getpid()}), flush=True)
        return

    import torch
    import torch.distributed as dist
    import dvslam_amd
    from dvslam_amd import synth
    from dvslam_amd import dist as dvdist

    assert torch.cuda.is_available(), "bench.py needs the MI355X (no CPU fallback)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if args.shard == "levels":
        return level_sharded_bench(args, world, rank, local, dev)
    rows, cols, B = 720, 1280, args.batch
    NB = 1 if args.single_resident_batch else max(1, args.resident_batches)
    # synthetic input, resident in HBM before the timed region: this rank's shard of NB global batches
    d_img, frames_distinct = make_batches(synth, torch, dev, B, NB, rank, rows, cols, not args.single_resident_batch)

    # One pipeline per GPU: extractor + matcher handles and their streams.  The match runs on the matrix cores (k_match_mfma)
    # and extraction on the vector ALUs, so the match of step i gets its OWN stream and runs beside the extraction of step
    # i + 1 instead of after its own: three output sets rotate (step i writes set i % 3, its match reads sets i % 3 and
    # (i - 1) % 3, and set i % 3 is not overwritten before step i + 3, by which time match i + 1 — its last reader — is
    # two steps old; the wait on it is stated anyway).  Every step still runs the whole path on one full batch.
    # The pipeline overlaps kernels on five streams (main, blur, next-batch pyramid, match, boundary exchange).  All are
    # created by the library, back to back, BEFORE RCCL comes up and none comes from torch's stream pool: HIP maps streams
    # to hardware queues in creation order, and with the exchange on a torch pool stream or RCCL initialised first the
    # same job ran anywhere between 48 k and 72 k frames/s depending on GPU_MAX_HW_QUEUES (kernels alone on the GPU took
    # 1.3-2.5x longer); like this it is 71 k for 4, 6, 8 and 12 queues.
    NP = 1
    NSETS = int(os.environ.get("BENCH_NSETS", "4"))   # output sets: set i % NSETS is overwritten by step i + NSETS; its last reader is match i + 1
    orb = dvslam_amd.ORBextractor(args.nfeatures, 1.2, 8, 20, 7, device=local, max_batch=B)
    # streams are created only when used: every HIP stream is a hardware queue, and one idle queue too many cost 0.2 ms per step
    # (measured: an unused fourth stream in the extractor handle, 0.74 -> 0.93 ms)
    need_x = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ) or os.environ.get("DVS_FORCE_COLLECTIVE") == "1"
    ts = torch.cuda.ExternalStream(orb.get_stream(), device=dev)
    ms = torch.cuda.ExternalStream(dvslam_amd.stream_create(local, int(os.environ.get("BENCH_M_PRIO", "0"))), device=dev) if (args.match_stream or args.match_late) else ts   # match
    # boundary exchange: with the pipelined match it shares the match stream (the match is its only consumer and a fifth hardware
    # queue cost 0.28 ms per step under the launcher); the serial schedule gives it a stream of its own beside the extraction
    xs = (ms if args.match_late else torch.cuda.ExternalStream(dvslam_amd.stream_create(local), device=dev)) if need_x else ts
    mat = dvslam_amd.BFMatcher(device=local, stream=ms.cuda_stream)
    cap = orb.capacity
    with torch.cuda.stream(ts):
        bufs = dict(kps=[torch.empty((B, cap, 28), dtype=torch.uint8, device=dev) for _ in range(NSETS)],
                    desc=[torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev) for _ in range(NSETS)],
                    n=[torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(NSETS)],
                    idx=torch.empty((B, cap), dtype=torch.int32, device=dev),
                    dist=torch.empty((B, cap), dtype=torch.int32, device=dev))
    P = dict(orb=orb, mat=mat, stream=ts, mstream=ms, xstream=xs, xdone=torch.cuda.Event(),
             ext_done=[torch.cuda.Event() for _ in range(NSETS)], match_done=[torch.cuda.Event() for _ in range(NSETS)], prev=None, **bufs)
    pipes = [P]
    torch.cuda.synchronize()
    # RCCL comes up AFTER the pipeline's handles and streams exist (see the stream comment above)
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ   # under torchrun (also with one rank): exercise RCCL
    comm = None
    rccl = None
    if world > 1 or launched:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
        dist.barrier()
        if not args.torch_exchange:
            def bcast_id(ident):   # the out-of-band hand-over of the RCCL unique id: torch.distributed's store
                box = [ident]
                dist.broadcast_object_list(box, src=0)
                return box[0]
            comm = dvdist.Comm(local, rank, world, bcast_id)
            rccl = {"nranks": world, "version": comm.rccl_version, "exchange": "dvs_exchange_boundary: ncclAllGather behind the C-ABI"}
            print(f"[bench] rank {rank}: RCCL communicator of {world} ranks (version {comm.rccl_version})", file=sys.stderr, flush=True)
        else:
            rccl = {"nranks": world, "exchange": "torch.distributed all_gather_into_tensor (dvslam_amd/dist.py)"}

    cap = pipes[0]["orb"].capacity
    torch.cuda.synchronize()
    state = {"i": 0}
    collective = world > 1 or launched or os.environ.get("DVS_FORCE_COLLECTIVE") == "1"
    if collective and comm is None and not dist.is_initialized():
        collective = False

    lag = 1 if args.match_late else 0    # step i extracts batch i and matches batch i - lag
    ev_fast = None
    if lag:
        ev_fast = torch.cuda.Event()
        ev_fast.record(ts)               # creates the hipEvent_t the library records behind FAST from now on
        orb.set_after_fast_event(ev_fast.cuda_event)

    defer = bool(args.defer == "on" and args.match_late and args.prefetch)
    # cross-stream joins cost a barrier packet each (5-8 us when they sit between two dependent kernels of one queue): the library
    # records the caller's output event itself (and gates its next prefetch on it) and takes the reuse guard onto the blur's
    # stream, so that nothing but FAST follows the previous step's descriptor kernel on the main stream.  BENCH_HOPS=0: the plain
    # stream-level statements of the same dependencies (A/B).
    lib_events = os.environ.get("BENCH_HOPS", "1") != "0" and ms is not ts
    if defer or lib_events:
        for e in P["ext_done"]:
            e.record(ts)                 # creates the hipEvent_t handles the library records where the outputs are complete

    def step_torch():
        i = state["i"]; state["i"] += 1
        s = i % NSETS
        img = d_img[i % NB]; nxt = d_img[(i + 1) % NB]
        T, M, X = P["stream"], P["mstream"], P["xstream"]
        j = i - lag                      # the batch matched in this step
        sj = j % NSETS
        if i >= NSETS and M is not T:
            guard = P["match_done"][(i - NSETS + 1) % NSETS]         # the last reader of the set this step overwrites
            if lib_events:
                P["orb"].set_reuse_guard_event(guard.cuda_event)
            else:
                T.wait_event(guard)
        with torch.cuda.stream(T):
            if args.prefetch:
                # streaming: the next batch is already resident, so its pyramid is built beside this batch's FAST
                # (every step still builds exactly one pyramid; the one of step 0 is built in-step)
                P["orb"].hint_next_batch_device(nxt.data_ptr())
            if defer or lib_events:
                P["orb"].set_output_event(P["ext_done"][s].cuda_event, defer=defer)   # recorded by the library: batch i's outputs complete
            P["orb"].extract_batch_device(img.data_ptr(), B, rows, cols, cols, rows * cols, P["kps"][s].data_ptr(),
                                          P["desc"][s].data_ptr(), cap, P["n"][s].data_ptr())
            if not (defer or lib_events):
                P["ext_done"][s].record(T)
        prev_desc = prev_n = 0
        if j >= 0 and collective:
            # the one exchange step, once per global batch: every rank's LAST frame of batch j; a rank's first frame is matched
            # against the frame before it in the global order — the previous rank's last frame of the same batch, or (rank 0) the
            # last rank's of the batch before.  It depends only on batch j's extraction and is joined before the match.
            qd, qn = P["desc"][sj][B - 1], P["n"][sj][B - 1:B]
            X.wait_event(P["ext_done"][sj])
            if comm is not None:
                prev_desc, prev_n = comm.exchange_boundary(X.cuda_stream, qd.data_ptr(), qn.data_ptr(), cap)
                P["xdone"].record(X)
            else:
                with torch.cuda.stream(X):
                    bd, bn = dvdist.exchange_boundary(qd, qn, cap)
                    P["xdone"].record(X)
                P["prev"] = (bd, bn)     # keep the gathered block alive until the match has read it
                prev_desc, prev_n = (bd.data_ptr(), bn.data_ptr()) if bd is not None else (0, 0)
        elif j > 0:
            sp = (j - 1) % NSETS         # one GPU: the previous batch's last frame, read in place
            prev_desc, prev_n = P["desc"][sp][B - 1].data_ptr(), P["n"][sp][B - 1:B].data_ptr()
        if j >= 0:
            if M is not T:
                M.wait_event(P["ext_done"][sj])
                if lag:
                    M.wait_event(ev_fast)    # recorded behind this step's FAST by the extraction just enqueued
            if collective:
                M.wait_event(P["xdone"])
            P["mat"].match_sequence_device(P["desc"][sj].data_ptr(), P["n"][sj].data_ptr(), cap, B, prev_desc, prev_n,
                                           P["idx"].data_ptr(), P["dist"].data_ptr())
            P["match_done"][sj].record(M)
        P["cur"] = s

    # everything a step needs as plain integers, prepared once: at 1-2 frames per step the host's enqueue time (torch tensor indexing,
    # data_ptr(), stream context managers: ~0.1 ms per step) was what the GPU waited for
    L_ = dvslam_amd.lib()
    ptr = dict(img=[d_img[k].data_ptr() for k in range(NB)],
               kps=[t.data_ptr() for t in P["kps"]], desc=[t.data_ptr() for t in P["desc"]], n=[t.data_ptr() for t in P["n"]],
               last_desc=[t[B - 1].data_ptr() for t in P["desc"]], last_n=[t[B - 1:B].data_ptr() for t in P["n"]],
               idx=P["idx"].data_ptr(), dist=P["dist"].data_ptr())
    T_, M_, X_ = P["stream"].cuda_stream, P["mstream"].cuda_stream, P["xstream"].cuda_stream
    raw = lib_events and not args.torch_exchange     # library events / waits by handle (torch events only on the fallback paths)
    if raw:
        for e in P["match_done"]:
            e.record(P["mstream"])
        evh = dict(ext=[e.cuda_event for e in P["ext_done"]], md=[e.cuda_event for e in P["match_done"]], fast=ev_fast.cuda_event if ev_fast else 0)
        P["xdone"].record(P["xstream"])
        evh["x"] = P["xdone"].cuda_event
    orb_h, mat_ = P["orb"], P["mat"]
    skip_match = os.environ.get("BENCH_NO_MATCH") == "1"   # diagnostics only: the extraction pipeline alone (the line is then NOT the metric)

    def step_raw():
        i = state["i"]; state["i"] += 1
        s = i % NSETS
        j = i - lag
        sj = j % NSETS
        if i >= NSETS:
            orb_h.set_reuse_guard_event(evh["md"][(i - NSETS + 1) % NSETS])   # the last reader of the set this step overwrites
        if args.prefetch:
            orb_h.hint_next_batch_device(ptr["img"][(i + 1) % NB])
        orb_h.set_output_event(evh["ext"][s], defer=None)
        orb_h.extract_batch_device(ptr["img"][i % NB], B, rows, cols, cols, rows * cols, ptr["kps"][s], ptr["desc"][s], cap, ptr["n"][s])
        prev_desc = prev_n = 0
        if j >= 0 and collective:
            L_.dvs_stream_wait_event(X_, evh["ext"][sj])
            prev_desc, prev_n = comm.exchange_boundary(X_, ptr["last_desc"][sj], ptr["last_n"][sj], cap)
            L_.dvs_event_record(evh["x"], X_)
        elif j > 0:
            sp = (j - 1) % NSETS
            prev_desc, prev_n = ptr["last_desc"][sp], ptr["last_n"][sp]
        if j >= 0:
            L_.dvs_stream_wait_event(M_, evh["ext"][sj])
            if lag:
                L_.dvs_stream_wait_event(M_, evh["fast"])
            if collective and X_ != M_:
                L_.dvs_stream_wait_event(M_, evh["x"])
            if not skip_match:
                mat_.match_sequence_device(ptr["desc"][sj], ptr["n"][sj], cap, B, prev_desc, prev_n, ptr["idx"], ptr["dist"])
            L_.dvs_event_record(evh["md"][sj], M_)
        P["cur"] = s

    if raw:
        orb_h.set_output_event(evh["ext"][0], defer=defer)
    step = step_raw if (raw and comm is not None or raw and not collective) else step_torch

    def sync_all():
        P["orb"].synchronize()
        P["xstream"].synchronize()
        P["stream"].synchronize()
        P["mstream"].synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    host_enqueue = time.perf_counter() - t0   # host time to enqueue all steps (no synchronisation inside)
    sync_all()
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # the pipelined match against the same job enqueued serially (outside the timed region): guards the event dependencies
    match_check = None; nj = None
    if world == 1 and state["i"] >= 3:   # (under the launcher with one rank this also checks the block that came back from RCCL)
        jl = state["i"] - 1 - lag
        sj, sp = jl % NSETS, (jl - 1) % NSETS
        idx2 = torch.empty_like(P["idx"]); dist2 = torch.empty_like(P["dist"])
        sync_all()
        P["mat"].match_sequence_device(P["desc"][sj].data_ptr(), P["n"][sj].data_ptr(), cap, B, P["desc"][sp][B - 1].data_ptr(),
                                       P["n"][sp][B - 1:B].data_ptr(), idx2.data_ptr(), dist2.data_ptr())
        sync_all()
        nj = P["n"][sj].cpu().numpy()
        same = all(bool(torch.equal(idx2[f, :nj[f]], P["idx"][f, :nj[f]]) and torch.equal(dist2[f, :nj[f]], P["dist"][f, :nj[f]])) for f in range(B))
        match_check = "identical to the serial match of the same batch" if same else "MISMATCH"
        if not same:
            print("[bench] WARNING: the pipelined match differs from the same job enqueued serially", file=sys.stderr, flush=True)
    if os.environ.get("DVS_DEBUG") and int(os.environ["DVS_DEBUG"]) & 4:   # diagnostics: when did the quad-tree workgroups of the last step run
        import numpy as np
        sync_all()
        st_, en_ = [], []
        for f in range(B):
            for l in range(8):
                o = np.zeros(64, np.uint64)
                dvslam_amd.lib().dvs_test_octree_stamps_frame(orb._h, f, l, o.ctypes.data)
                c = int(o[0])
                if c >= 2:
                    st_.append((int(o[1]) & 0xFFFFFFFFFFFFFF, l, f)); en_.append((int(o[c]) & 0xFFFFFFFFFFFFFF, l, f))
        t0 = min(x[0] for x in st_)
        per_level = {l: (max((x[0] - t0) / 100 for x in st_ if x[1] == l), max((x[0] - t0) / 100 for x in en_ if x[1] == l)) for l in range(8)}
        print("[bench] quad-tree workgroups of the last step: latest start / latest end per level (us after the first start):",
              {l: (round(a, 1), round(b, 1)) for l, (a, b) in per_level.items()}, file=sys.stderr, flush=True)
        late = sorted(((x[0] - t0) / 100 for x in st_), reverse=True)[:10]
        print("[bench] ten latest starts:", [round(x, 1) for x in late], file=sys.stderr, flush=True)
    # per-kernel durations: K more steps on ONE pipeline with hipEvents around every stage launch (the events cost ~10 us
    # of stream time per stage, so they stay out of the whole-job timing above)
    # (a) the same schedule as the timed region (overlap + pyramid prefetch), events on the streams the kernels run on: what a
    #     kernel takes WHILE its neighbours share the machine (rocprofv3's kernel statistics of this command show these)
    orb.enable_stage_timing(True)
    with torch.cuda.stream(P["stream"]):
        for k in range(args.steps):
            if defer or lib_events:
                orb.set_output_event(P["ext_done"][k % NSETS].cuda_event, defer=defer)
            if args.prefetch:
                orb.hint_next_batch_device(d_img[(k + 1) % NB].data_ptr())
            orb.extract_batch_device(d_img[k % NB].data_ptr(), B, rows, cols, cols, rows * cols, P["kps"][P["cur"]].data_ptr(),
                                     P["desc"][P["cur"]].data_ptr(), cap, P["n"][P["cur"]].data_ptr())
    sync_all()
    ov_ms, ov_calls = orb.stage_times()
    orb.enable_stage_timing(False)
    # (b) every kernel alone on the stream: the kernel's own duration, which the rooflines below are computed from
    orb.set_output_event(0, defer=False)
    orb.set_overlap(False)
    orb.enable_stage_timing(True)
    with torch.cuda.stream(P["stream"]):
        for k in range(args.steps):
            orb.extract_batch_device(d_img[k % NB].data_ptr(), B, rows, cols, cols, rows * cols, P["kps"][P["cur"]].data_ptr(),
                                     P["desc"][P["cur"]].data_ptr(), cap, P["n"][P["cur"]].data_ptr())
    sync_all()
    stage_ms, stage_calls = orb.stage_times()
    orb.enable_stage_timing(False)
    orb.set_overlap(True)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    d_n = pipes[0]["n"][pipes[0]["cur"]]; d_dist = pipes[0]["dist"]

    n_host = d_n.cpu().numpy()
    matched = int((d_dist[:, :].cpu().numpy()[1, :(nj if nj is not None else n_host)[1]] < 50).sum()) if B > 1 else 0

    if rank == 0:
        total_frames = world * B * args.steps
        fps = total_frames / elapsed
        dom = max(stage_ms, key=lambda k: stage_ms[k])
        traffic = None   # HBM bytes per launch from the committed PMC passes (profiles/pmc_traffic.json), scaled to this batch
        issue = None     # wave-level VALU instructions per second of the dominant kernel against the chip's issue rates
        dom_ms = stage_ms[dom] / max(stage_calls[dom], 1)
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            traffic = int(tj["bytes_per_launch"][dom] * B / tj["batch"])
            insts = tj["valu_insts_per_launch"][dom] * B / tj["batch"]
            rate = insts / (dom_ms * 1e-3) / 1e9
            nominal = 256 * 4 * 2.4 / 2   # 1 024 SIMD-32 x one wave64 instruction per 2 cycles at 2.4 GHz (MI355X_MICROARCH.md)
            issue = {"kernel": dom, "valu_wave_insts_per_launch": int(insts), "achieved": round(rate, 1), "peak": nominal,
                     "unit": "G wave-instr/s", "frac": round(rate / nominal, 4),
                     "measured_class_rates_G_per_s": {"add/xor/max_i16 class": "780-950", "perm/pk16/dot/bcnt/min3/mad24 class": tj["valu_issue_peak_G_per_s"]},
                     "source": "SQ_INSTS_VALU from profiles/pmc_traffic.json; class rates: tools/ubench/valu_rate.hip, profiles/r02_valu_issue_rates.txt"}
        except Exception:
            pass
        achieved = STAGE_BYTES[dom] * B / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        out = {
            "metric": "frames/sec ORB+match @1280x720x2000kp", "value": round(fps, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "1280x720 gray frames, ORBextractor(2000,1.2,8,20,7) extract + BFMatcher(HAMMING) match vs previous frame "
                                   "(BASELINE configs[1])", "frames_per_gpu_per_step": B, "frames_distinct": frames_distinct,
                       "resident_batches": NB, "keypoints_frame1": int(n_host[min(1, B - 1)]),
                       "matches_lt50_frame1": matched, "parallelism": f"frame-sharded x{world}, boundary-descriptor all_gather, match of batch i on its own stream beside the extraction "
                                      f"of batch i + 1" if args.match_stream else (f"frame-sharded x{world}, boundary-descriptor all_gather, step i = extraction of batch i + match of batch i - 1 "
                                      f"released behind FAST{', descriptor stage of batch i beside FAST of batch i + 1' if defer else ''}" if args.match_late else f"frame-sharded x{world}, boundary-descriptor all_gather, serial match")},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": round(achieved * 1e9 / HBM_PEAK, 5), "traffic": traffic,
                         "ms_per_launch": round(dom_ms, 4), "algorithmic_bytes_per_launch": STAGE_BYTES[dom] * B},
            "valu_issue_roofline": issue,
            "hbm_read_roofline_frac": round(fps / world * READ_BYTES_PER_FRAME / HBM_PEAK, 5),
            "stage_ms_per_launch_isolated": {k: round(v / max(stage_calls[k], 1), 4) for k, v in stage_ms.items()},
            "stage_ms_per_launch_overlapped": {k: round(v / max(ov_calls[k], 1), 4) for k, v in ov_ms.items()},
            "host_enqueue_ms_per_step": round(host_enqueue / args.steps * 1e3, 4), "pipelines_per_gpu": NP, "rccl": rccl, "match_check": match_check,
        }
        if world == 1 and not args.no_cpu_baseline:
            import oracle_bindings as ob
            ob.use_native()   # -O2 -march=native, built here on the host that times it
            cb_frames = [synth.make_frame(t, cols, rows) for t in range(min(48, 64))]
            out["cpu_baseline"] = cpu_baseline(cb_frames, args.nfeatures)
            out["speedup_vs_cpu_1thread"] = round(fps / out["cpu_baseline"]["value"], 1)
            nthr = max(1, min(16, len(os.sched_getaffinity(0))))   # the GPU box's CPU share for one GPU
            out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(cb_frames, args.nfeatures, nthr)
            out["speedup_vs_cpu_all_cores"] = round(fps / out["cpu_baseline_all_cores"]["value"], 1)
            out["ba"] = ba_bench(dvslam_amd, synth, local)
        print(json.dumps(out), flush=True)
    if comm is not None:
        torch.cuda.synchronize()
        comm.close()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
