#!/usr/bin/env python3
"""Experiment: NP independent extractor+matcher pipelines on one GPU, step i on pipeline i % NP (each with its own streams and
buffers), so that one batch's FAST can fill the issue slots another batch's quad-tree / descriptor / match stages leave idle.
usage: python tools/exp_two_pipes.py NP [steps]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dynamic-visual-slam_amd")); sys.path.insert(0, ROOT)
import torch
import dvslam_amd
from dvslam_amd import synth
import bench

NPIPE = int(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 120
dev = torch.device("cuda", 0); rows, cols, B, NB = 720, 1280, 64, 6
d_img, _ = bench.make_batches(synth, torch, dev, B, NB, 0, rows, cols, True)
pipes = []
for p in range(NPIPE):
    orb = dvslam_amd.ORBextractor(2000, 1.2, 8, 20, 7, device=0, max_batch=B)
    ts = torch.cuda.ExternalStream(orb.get_stream(), device=dev)
    mat = dvslam_amd.BFMatcher(device=0, stream=ts.cuda_stream)
    cap = orb.capacity
    with torch.cuda.stream(ts):
        pipes.append(dict(orb=orb, mat=mat, ts=ts, kps=torch.empty((B, cap, 28), dtype=torch.uint8, device=dev),
                          desc=torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev), n=torch.zeros(B, dtype=torch.int32, device=dev),
                          idx=torch.empty((B, cap), dtype=torch.int32, device=dev), dist=torch.empty((B, cap), dtype=torch.int32, device=dev),
                          ext=torch.cuda.Event(), done=torch.cuda.Event()))
torch.cuda.synchronize()
state = {"i": 0}

def step():
    i = state["i"]; state["i"] += 1
    P = pipes[i % NPIPE]; Q = pipes[(i - 1) % NPIPE]
    with torch.cuda.stream(P["ts"]):
        P["orb"].hint_next_batch_device(d_img[(i + NPIPE) % NB].data_ptr())
        P["orb"].extract_batch_device(d_img[i % NB].data_ptr(), B, rows, cols, cols, rows * cols, P["kps"].data_ptr(), P["desc"].data_ptr(), cap, P["n"].data_ptr())
        P["ext"].record(P["ts"])
    pd = pn = 0
    if i > 0:
        if Q is not P:
            P["ts"].wait_event(Q["ext"])      # the previous batch's last frame (another pipeline's output)
        pd, pn = Q["desc"][B - 1].data_ptr(), Q["n"][B - 1:].data_ptr()
    # NOTE with NPIPE == 1 the previous batch's block is overwritten by this extraction; the experiment only measures time
    P["mat"].match_sequence_device(P["desc"].data_ptr(), P["n"].data_ptr(), cap, B, pd, pn, P["idx"].data_ptr(), P["dist"].data_ptr())
    P["done"].record(P["ts"])

for _ in range(12): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps): step()
torch.cuda.synchronize()
el = time.perf_counter() - t0
print(json.dumps({"pipes": NPIPE, "ms_per_step": round(el / steps * 1e3, 4), "frames_per_s": round(B * steps / el, 1)}))
