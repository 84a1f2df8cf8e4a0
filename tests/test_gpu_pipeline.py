"""The configuration bench.py times, under the oracle.

* test_timed_configuration_against_oracle: dvslam_amd/pipeline.py's step — the one bench.py runs — at BASELINE configs[1] in the
  bench's own shape: 64 frames of 1280x720 per step, 2000 keypoints, software-pipelined match, deferred descriptor stage, reuse
  guard, 4 output sets, several steps over distinct batches.  EVERY frame's keypoints / descriptors and EVERY match job are compared
  with the CPU oracle (bit-exact).
* test_frame_sharded_loopback_equals_single_rank: BASELINE configs[3] on one GPU — 8 logical ranks x 8 frames, each rank its own
  extractor / matcher / streams and host thread, dvs_exchange_boundary once per global batch through the loopback communicator
  (csrc/comm.hip: the multi-rank branches — previous rank's block, wrap-around to the last rank's block of the previous call — run
  here); the 64 match jobs of every global batch equal the single-rank 64-frame sequence and the oracle.
"""
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest
from dvslam_amd import synth

pytestmark = pytest.mark.gpu
ROWS, COLS, NF = 720, 1280, 2000


def _global_batch(g, n=64):
    """global batch g: n consecutive frames of the synthetic sequence over a scene of its own (bench.py's make_batches)"""
    out = np.stack([synth.make_frame(i, COLS, ROWS, seed=1234 + 101 * g) for i in range(n)])
    synth._CANVAS_CACHE.clear()
    return out


def _oracle_extract_all(oracle, frames, threads=8):
    """oracle results of every frame, one extractor instance per worker thread (ctypes releases the GIL inside the C++ calls)"""
    local = threading.local()

    def one(img):
        if not hasattr(local, "o"):
            local.o = oracle.OracleORB(NF, 1.2, 8, 20, 7)
        return local.o.extract(img)
    with ThreadPoolExecutor(threads) as ex:
        return list(ex.map(one, frames))


def _assert_frame(got, ref, where):
    (n, k, d), (n2, k2, d2) = got, ref
    assert int(n) == n2, (where, int(n), n2)
    assert k[:n2].tobytes() == k2.tobytes(), where          # all 7 cv::KeyPoint fields, float bit patterns included
    assert (d[:n2] == d2).all(), where


def test_timed_configuration_against_oracle(gpu, oracle):
    from dvslam_amd import _lib
    from dvslam_amd.pipeline import StreamingPipeline
    B, NB, STEPS = 64, 2, 5
    batches = [_global_batch(g, B) for g in range(NB)]
    ref = [_oracle_extract_all(oracle, b) for b in batches]              # [batch][frame] = (n, kps, desc)
    d_img = [_lib.DeviceBuffer(b.nbytes).upload(b) for b in batches]
    pipe = StreamingPipeline(B, ROWS, COLS, NF, nsets=4, pipelined=True)   # bench.py's construction
    for i in range(STEPS):
        pipe.step(d_img[i % NB].ptr, d_img[(i + 1) % NB].ptr)              # bench.py's step()
    pipe.flush()
    pipe.synchronize()
    for i in range(STEPS - pipe.nsets, STEPS):                             # the batches still resident: steps 1 .. 4
        n, k, d = pipe.outputs(i)
        for f in range(B):
            _assert_frame((n[f], k[f], d[f]), ref[i % NB][f], f"step {i} frame {f}")
    checked = 0
    for j in range(STEPS - pipe.nsets, STEPS):                             # every match job of those batches, frame 0 against the batch before
        idx, dist = pipe.matches(j)
        for f in range(B):
            q = ref[j % NB][f]
            t = ref[j % NB][f - 1] if f else ref[(j - 1) % NB][B - 1]
            i2, d2 = oracle.match(q[2], t[2])
            assert (idx[f, :q[0]] == i2).all() and (dist[f, :q[0]] == d2).all(), f"match job {f} of batch {j}"
            checked += 1
    assert checked == 4 * B
    pipe.close()


@pytest.mark.parametrize("B,rows,cols,nf,nsets,steps", [(5, 480, 640, 800, 3, 8), (33, 360, 1000, 700, 3, 5), (1, 720, 1280, 2000, 4, 9), (9, 250, 332, 200, 5, 6),
                                                        (16, 480, 640, 600, 3, 4), (24, 360, 1000, 500, 3, 4)])
def test_other_shapes_of_the_pipelined_step(gpu, oracle, B, rows, cols, nf, nsets, steps):
    """the same step at other batch sizes (1 and 33: the small-batch kernel variants and the 256-thread quad-tree on either side of
    their thresholds; 16 and 24: the pyramid kernel's lanes run over 2 / 3 frames of an XCD's share, 64 in the test above over 8), resolutions (widths that are not multiples of 4 take the generic kernels) and output-set counts: the resident
    batches and their match jobs against the oracle"""
    _run_shape(oracle, B, rows, cols, nf, nsets, steps)


def test_pipelined_step_with_the_lds_free_matrix_core_blur(gpu, oracle, monkeypatch):
    """DVS_BLUR_MFMA=2 (k_blur_mfma_direct) in the pipelined schedule"""
    monkeypatch.setenv("DVS_BLUR_MFMA", "2")
    _run_shape(oracle, 5, 480, 640, 800, 3, 8)


@pytest.mark.parametrize("B,lanes,nsets,steps", [(1, 4, 4, 11), (2, 2, 4, 9), (8, 4, 4, 7), (8, 1, 3, 5), (3, 3, 3, 8), (16, 2, 6, 8), (8, 3, 0, 11), (1, 0, 0, 13)])
def test_lane_schedule_for_small_batches(gpu, oracle, B, lanes, nsets, steps):
    """the small-batch schedule (dvs_pipeline_params::lanes): whole steps in flight on `lanes` extractor / matcher pairs, cross-lane
    order by events only — every resident frame and match job against the oracle, at 1280x720 / 2000 like the timed configuration;
    (8, 1): the two-stream software pipeline forced at a lane-sized batch"""
    p = _run_shape(oracle, B, 720, 1280, 2000, nsets, steps, lanes=lanes)
    assert p == (lanes or 4)                                              # lanes = 0: four lanes (and eight sets) up to 4 frames per step


@pytest.mark.parametrize("B,rows,cols,nf,nsets,steps", [(8, 720, 1280, 2000, 4, 7), (12, 480, 640, 800, 3, 7), (40, 360, 1000, 700, 4, 6), (7, 250, 332, 200, 3, 6)])
def test_four_stream_schedule(gpu, oracle, B, rows, cols, nf, nsets, steps):
    """dvs_pipeline_params::quadtree_async = 1: the quad-tree on the auxiliary stream beside the next FAST (three candidate sets), the
    descriptor stage on the match stream, the blur on the main stream ahead of FAST (three blurred blocks).  40 frames: more than one
    quad-tree workgroup per CU, launched per level class with graded LDS; 332 columns: rows that are not dword aligned keep the plain
    path although the switch is on"""
    _run_shape(oracle, B, rows, cols, nf, nsets, steps, lanes=1, quadtree_async=1)


@pytest.mark.parametrize("B,rows,cols,nf,qa", [(8, 720, 1280, 2000, 1), (6, 480, 640, 600, -1), (20, 360, 500, 400, -1)])
def test_level_chain_as_graph_launch(gpu, oracle, monkeypatch, B, rows, cols, nf, qa):
    """the announced batch's level chain enqueued as one hipGraphLaunch once its argument set (source block, frame count, destination
    pyramid) has come back (dvs_orb_chain_graph_launches): the bytes of the steps that ran on graphs equal the oracle's.  Automatic up to
    12 frames per step (first two cases: four-stream and two-stream form); forced on for 20."""
    if B > 12:
        monkeypatch.setenv("DVS_CHAIN_GRAPH", "1")
    graphs = []
    _run_shape(oracle, B, rows, cols, nf, 4, 22, lanes=1, quadtree_async=qa, hooks=True, probe=lambda pipe: graphs.append(
        _lib_mod().test_lib().dvs_orb_chain_graph_launches(pipe.orb._h)))
    assert graphs[0] >= 8, graphs                                      # 3 resident batches x 2 or 3 pyramids: every set seen twice by step 13
    monkeypatch.setenv("DVS_CHAIN_GRAPH", "0")
    graphs.clear()
    _run_shape(oracle, B, rows, cols, nf, 4, 5, lanes=1, quadtree_async=qa, hooks=True, probe=lambda pipe: graphs.append(
        _lib_mod().test_lib().dvs_orb_chain_graph_launches(pipe.orb._h)))
    assert graphs == [0]


def test_four_stream_is_the_default_from_7_to_24_frames(gpu, oracle):
    """DVS_PIPELINE_LANE_BATCH < batch <= DVS_PIPELINE_ASYNC_BATCH (6, 24): lanes below, the two-stream pipeline above; one default shape
    in the upper half of the range against the oracle (rings of four sets: 9 steps go round twice)"""
    from dvslam_amd.pipeline import StreamingPipeline
    for B, want_lanes, want_async in [(4, 4, False), (6, 4, False), (7, 1, True), (24, 1, True), (25, 1, False)]:
        pipe = StreamingPipeline(B, 240, 320, 300, nsets=0)
        assert (pipe.lanes, pipe.quadtree_async) == (want_lanes, want_async), B
        pipe.close()
    _run_shape(oracle, 18, 480, 640, 700, 4, 9)


def test_level_chain_graph_cache_stays_bounded(gpu, oracle):
    """a caller streaming from more buffers than the graph cache holds (17 blocks x 4 pyramids > 48 argument sets): the cache is dropped
    and refilled on the way, the results stay the oracle's"""
    from dvslam_amd import _lib
    from dvslam_amd.pipeline import StreamingPipeline
    B, rows, cols, nf, NB, steps = 7, 240, 320, 300, 17, 150
    frames = [np.stack([synth.make_frame(3 * g + i, cols, rows, seed=5 + g) for i in range(B)]) for g in range(NB)]
    d_img = [_lib.DeviceBuffer(b.nbytes).upload(b) for b in frames]
    pipe = StreamingPipeline(B, rows, cols, nf, nsets=4, pipelined=True, hooks=True)
    assert pipe.quadtree_async
    # 45 steps over 5 blocks (20 argument sets: graphs from the second round on), then all 17 (the 49th set drops the cache; with 68 sets
    # in rotation none comes back before the next drop: plain launches from there on)
    order = [i % 5 for i in range(45)] + [i % NB for i in range(steps - 45)]
    for i in range(steps):
        pipe.step(d_img[order[i]].ptr, d_img[order[i + 1]].ptr if i + 1 < steps else 0)
    pipe.flush(); pipe.synchronize()
    assert 15 <= _lib.test_lib().dvs_orb_chain_graph_launches(pipe.orb._h) <= 45
    o = oracle.OracleORB(nf, 1.2, 8, 20, 7)
    for i in range(steps - 3, steps):
        n, k, d = pipe.outputs(i)
        for f in (0, B - 1):
            _assert_frame((n[f], k[f], d[f]), o.extract(frames[order[i]][f]), f"step {i} frame {f}")
    pipe.close()


def _lib_mod():
    from dvslam_amd import _lib
    return _lib


def _run_shape(oracle, B, rows, cols, nf, nsets, steps, lanes=0, quadtree_async=0, probe=None, hooks=False):
    from dvslam_amd import _lib
    from dvslam_amd.pipeline import StreamingPipeline
    NB = 3
    frames = [np.stack([synth.make_frame(7 * g + i, cols, rows, seed=77 + g) for i in range(B)]) for g in range(NB)]
    o = oracle.OracleORB(nf, 1.2, 8, 20, 7)
    ref = [[o.extract(f) for f in batch] for batch in frames]
    d_img = [_lib.DeviceBuffer(b.nbytes).upload(b) for b in frames]
    pipe = StreamingPipeline(B, rows, cols, nf, nsets=nsets, pipelined=True, lanes=lanes, quadtree_async=quadtree_async, hooks=hooks)
    assert quadtree_async == 0 or pipe.quadtree_async == (quadtree_async == 1)
    used, nsets = pipe.lanes, pipe.nsets
    for i in range(steps):
        pipe.step(d_img[i % NB].ptr, d_img[(i + 1) % NB].ptr)
    pipe.flush(); pipe.synchronize()
    for i in range(steps - nsets, steps):
        n, k, d = pipe.outputs(i)
        idx, dist = pipe.matches(i)
        for f in range(B):
            _assert_frame((n[f], k[f], d[f]), ref[i % NB][f], f"step {i} frame {f}")
            if i == 0 and f == 0:
                continue
            t = ref[i % NB][f - 1] if f else ref[(i - 1) % NB][B - 1]
            i2, d2 = oracle.match(ref[i % NB][f][2], t[2])
            assert (idx[f, :n[f]] == i2).all() and (dist[f, :n[f]] == d2).all(), (i, f)
    if probe:
        probe(pipe)
    pipe.close()
    return used


def test_serial_schedule_against_oracle(gpu, oracle):
    """bench.py --serial-match: the same step without the software pipeline (one stream, match behind its own extraction)"""
    from dvslam_amd import _lib
    from dvslam_amd.pipeline import StreamingPipeline
    B = 6
    frames = _global_batch(3, 2 * B)
    ref = _oracle_extract_all(oracle, frames)
    d_img = [_lib.DeviceBuffer(frames[:B].nbytes).upload(frames[:B]), _lib.DeviceBuffer(frames[B:].nbytes).upload(frames[B:])]
    pipe = StreamingPipeline(B, ROWS, COLS, NF, nsets=2, pipelined=False)
    pipe.step(d_img[0].ptr); pipe.step(d_img[1].ptr)
    pipe.synchronize()
    for i in range(2):
        n, k, d = pipe.outputs(i)
        idx, dist = pipe.matches(i)
        for f in range(B):
            t = i * B + f
            _assert_frame((n[f], k[f], d[f]), ref[t], f"frame {t}")
            if t:
                i2, d2 = oracle.match(ref[t][2], ref[t - 1][2])
                assert (idx[f, :ref[t][0]] == i2).all() and (dist[f, :ref[t][0]] == d2).all(), t
    pipe.close()


def test_frame_sharded_loopback_equals_single_rank(gpu, oracle):
    from dvslam_amd import _lib
    from dvslam_amd import dist as dvdist
    from dvslam_amd.pipeline import StreamingPipeline
    WORLD, B, NG, STEPS = 8, 8, 2, 4                                      # 8 ranks x 8 frames = global batches of 64, 4 of them (2 distinct)
    G = WORLD * B
    gb = [_global_batch(10 + g, G) for g in range(NG)]
    # single rank, 64 frames per step: the unsharded sequence
    one = StreamingPipeline(G, ROWS, COLS, NF, nsets=STEPS, pipelined=True)
    d_all = [_lib.DeviceBuffer(b.nbytes).upload(b) for b in gb]
    for i in range(STEPS):
        one.step(d_all[i % NG].ptr, d_all[(i + 1) % NG].ptr)
    one.flush(); one.synchronize()
    want = [(one.outputs(i), one.matches(i)) for i in range(STEPS)]
    one.close()
    d_rank = [[_lib.DeviceBuffer(gb[g][r * B:(r + 1) * B].nbytes).upload(gb[g][r * B:(r + 1) * B]) for g in range(NG)] for r in range(WORLD)]
    for lanes in (1, 2, 4):    # the two-stream software pipeline per rank; the lane schedule (4: with the priority-stream lane) (strong-scaling form: 8 frames per rank), whose
        _loopback_ranks(oracle, want, d_rank, WORLD, B, NG, STEPS, lanes)    # successive exchanges run on different streams


def _loopback_ranks(oracle, want, d_rank, WORLD, B, NG, STEPS, lanes):
    from dvslam_amd import dist as dvdist
    from dvslam_amd.pipeline import StreamingPipeline
    G = WORLD * B
    # 8 logical ranks, each with its own handles, streams, communicator and host thread
    pipes = [StreamingPipeline(B, ROWS, COLS, NF, nsets=STEPS, pipelined=True, lanes=lanes) for _ in range(WORLD)]
    assert all(p.lanes == lanes for p in pipes)
    comms = dvdist.Comm.loopback(0, WORLD)
    for r in range(WORLD):
        assert comms[r].rank == r and comms[r].world == WORLD
        pipes[r].attach_comm(comms[r])
    errors = []

    def run(r):
        try:
            for i in range(STEPS):
                pipes[r].step(d_rank[r][i % NG].ptr, d_rank[r][(i + 1) % NG].ptr)
            pipes[r].flush()
            pipes[r].synchronize()
        except Exception as e:   # noqa: BLE001
            errors.append((r, repr(e)))
    th = [threading.Thread(target=run, args=(r,)) for r in range(WORLD)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    jobs = 0
    for i in range(STEPS):
        (n1, k1, d1), (idx1, dist1) = want[i]
        for r in range(WORLD):
            n, k, d = pipes[r].outputs(i)
            idx, dist = pipes[r].matches(i)
            for f in range(B):
                gf = r * B + f
                assert n[f] == n1[gf] and k[f, :n[f]].tobytes() == k1[gf, :n[f]].tobytes() and (d[f, :n[f]] == d1[gf, :n[f]]).all(), (i, r, f)
                if i == 0 and gf == 0:
                    continue                                              # the very first frame has no predecessor
                assert (idx[f, :n[f]] == idx1[gf, :n[f]]).all() and (dist[f, :n[f]] == dist1[gf, :n[f]]).all(), f"match: batch {i} rank {r} frame {f}"
                # ... and the oracle's matcher on the same descriptors (shard boundaries included: f == 0 reads the exchanged block)
                t_desc, t_n = (d1[gf - 1], n1[gf - 1]) if gf else (want[i - 1][0][2][G - 1], want[i - 1][0][0][G - 1])
                i2, d2 = oracle.match(d[f, :n[f]], t_desc[:t_n])
                assert (idx[f, :n[f]] == i2).all() and (dist[f, :n[f]] == d2).all(), (i, r, f)
                jobs += 1
    assert jobs == STEPS * G - 1
    for p in pipes:
        p.close()
    for c in comms:
        c.close()
