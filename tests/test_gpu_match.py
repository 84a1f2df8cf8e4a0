"""GPU parity: HIP Hamming matcher (C-ABI) vs the oracle restatement of cv::BFMatcher::match — bit exact."""
import os
import numpy as np
import pytest
from dvslam_amd import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("nq,nt", [(1, 1), (1, 700), (63, 64), (64, 3), (65, 129), (300, 257), (2000, 2000), (2024, 1999), (5000, 4097)])
def test_match_parity(gpu, oracle, nq, nt):
    from dvslam_amd import BFMatcher
    q = synth.make_descriptors(nq, 100 + nq); t = synth.make_descriptors(nt, 200 + nt)
    if nt > 10 and nq > 3:                      # exact duplicates and near-duplicates: ties -> lowest train index
        t[7] = q[2]; t[3] = q[2]; t[nt - 1] = q[2]
        t[5] = q[1]; t[5, 0] ^= 1; t[9] = q[1]; t[9, 31] ^= 128
    m = BFMatcher()
    idx, d = m.match(q, t)
    idx2, d2 = oracle.match(q, t)
    assert (idx == idx2).all() and (d == d2).all()


def test_golden_fixture(gpu):
    from dvslam_amd import BFMatcher
    g = np.load(os.path.join(GOLD, "match_300x257.npz"))
    idx, d = BFMatcher().match(g["q"], g["t"])
    assert (idx == g["idx"]).all() and (d == g["dist"]).all()
    assert idx[3] == 5 and d[3] == 0            # duplicate rows 5 and 100 -> lowest index


def test_empty_inputs(gpu):
    from dvslam_amd import BFMatcher
    m = BFMatcher()
    q = synth.make_descriptors(10, 1)
    idx, d = m.match(q, np.zeros((0, 32), np.uint8))
    assert len(idx) == 0                         # empty train -> empty result (cv::BFMatcher)
    idx, d = m.match(np.zeros((0, 32), np.uint8), q)
    assert len(idx) == 0


def test_properties_full_size(gpu):
    from dvslam_amd import BFMatcher
    m = BFMatcher()
    q = synth.make_descriptors(2000, 11)
    idx, d = m.match(q, q)                       # self match: identity, distance 0
    assert (idx == np.arange(2000)).all() and (d == 0).all()
    perm = np.random.default_rng(3).permutation(2000)
    idx, d = m.match(q, q[perm])                 # permuted train: inverse permutation
    assert (perm[idx] == np.arange(2000)).all() and (d == 0).all()
    t = synth.make_descriptors(2000, 12)
    idx, d = m.match(q, t)
    x = np.unpackbits(q[:50, None, :] ^ t[None, :, :], axis=2).sum(axis=2)
    assert (d[:50] == x.min(axis=1)).all() and (idx[:50] == x.argmin(axis=1)).all()


@pytest.mark.parametrize("nq,nt,thr", [(1, 1, 50), (40, 3000, 100), (500, 700, 110), (300, 300, 257)])
def test_thresh_parity(gpu, oracle, nq, nt, thr):
    """backend association shape (backend.cpp:1068-1077): all pairs with distance < thr"""
    from dvslam_amd import BFMatcher
    q = synth.make_descriptors(nq, 5); t = synth.make_descriptors(nt, 6)
    t[0] = q[0]
    n, pairs = BFMatcher().match_thresh(q, t, thr)
    n2, pairs2 = oracle.match_thresh(q, t, thr)
    assert n == n2 and (pairs == pairs2).all()


def test_batch_device(gpu, oracle):
    from dvslam_amd import BFMatcher
    from dvslam_amd._lib import DeviceBuffer
    m = BFMatcher()
    P, S = 4, 520
    nq = np.array([500, 520, 1, 333], np.int32); nt = np.array([520, 17, 400, 333], np.int32)
    Q = np.stack([synth.make_descriptors(S, 30 + p) for p in range(P)]); T = np.stack([synth.make_descriptors(S, 40 + p) for p in range(P)])
    dq = DeviceBuffer(Q.nbytes).upload(Q); dt = DeviceBuffer(T.nbytes).upload(T)
    dnq = DeviceBuffer(16).upload(nq); dnt = DeviceBuffer(16).upload(nt)
    di = DeviceBuffer(P * S * 4); dd = DeviceBuffer(P * S * 4)
    m.match_batch_device(dq.ptr, dnq.ptr, S, dt.ptr, dnt.ptr, S, P, di.ptr, dd.ptr)
    m.synchronize()
    idx = di.download(np.int32, P * S).reshape(P, S); d = dd.download(np.int32, P * S).reshape(P, S)
    for p in range(P):
        i2, d2 = oracle.match(Q[p, :nq[p]], T[p, :nt[p]])
        assert (idx[p, :nq[p]] == i2).all() and (d[p, :nq[p]] == d2).all()


def test_sequence_device(gpu, oracle):
    """frame p vs frame p-1 in place, frame 0 vs a predecessor that lives in another buffer (or nothing)"""
    from dvslam_amd import BFMatcher
    from dvslam_amd._lib import DeviceBuffer
    m = BFMatcher()
    F, S = 5, 300
    n = np.array([300, 257, 0, 64, 299], np.int32)
    D = np.stack([synth.make_descriptors(S, 70 + p) for p in range(F)])
    D[1, 3] = D[0, 9]; D[1, 4] = D[0, 9]                      # exact matches / ties across consecutive frames
    prev = synth.make_descriptors(S, 99); nprev = np.array([211], np.int32)
    dd = DeviceBuffer(D.nbytes).upload(D); dn = DeviceBuffer(F * 4).upload(n)
    dp = DeviceBuffer(prev.nbytes).upload(prev); dnp = DeviceBuffer(4).upload(nprev)
    di = DeviceBuffer(F * S * 4); dx = DeviceBuffer(F * S * 4)
    for with_prev in (True, False):
        m.match_sequence_device(dd.ptr, dn.ptr, S, F, dp.ptr if with_prev else 0, dnp.ptr if with_prev else 0, di.ptr, dx.ptr)
        m.synchronize()
        idx = di.download(np.int32, F * S).reshape(F, S); d = dx.download(np.int32, F * S).reshape(F, S)
        for p in range(F):
            t = (prev[:nprev[0]] if with_prev else prev[:0]) if p == 0 else D[p - 1, :n[p - 1]]
            i2, d2 = oracle.match(D[p, :n[p]], t)
            assert (idx[p, :n[p]] == i2).all() and (d[p, :n[p]] == d2).all(), (with_prev, p)


def test_sequence_device_predecessor_larger_than_the_stride(gpu, oracle):
    """k_match_lds stages a job's train set in LDS sized by the declared row stride; the predecessor block of a sequence has no declared
    size — one with more rows than the stride is read from global memory instead (job 0 only), tie-heavy sets throughout"""
    from dvslam_amd import BFMatcher
    from dvslam_amd._lib import DeviceBuffer
    m = BFMatcher()
    F, S, NP = 3, 256, 450
    n = np.array([256, 200, 255], np.int32)
    D = np.stack([_tie_heavy(S, 170 + p) for p in range(F)])
    prev = _tie_heavy(NP, 199); nprev = np.array([NP], np.int32)
    dd = DeviceBuffer(D.nbytes).upload(D); dn = DeviceBuffer(F * 4).upload(n)
    dp = DeviceBuffer(prev.nbytes).upload(prev); dnp = DeviceBuffer(4).upload(nprev)
    di = DeviceBuffer(F * S * 4); dx = DeviceBuffer(F * S * 4)
    m.match_sequence_device(dd.ptr, dn.ptr, S, F, dp.ptr, dnp.ptr, di.ptr, dx.ptr)
    m.synchronize()
    idx = di.download(np.int32, F * S).reshape(F, S); d = dx.download(np.int32, F * S).reshape(F, S)
    for p in range(F):
        t = prev if p == 0 else D[p - 1, :n[p - 1]]
        i2, d2 = oracle.match(D[p, :n[p]], t)
        assert (idx[p, :n[p]] == i2).all() and (d[p, :n[p]] == d2).all(), p


def _tie_heavy(n, seed, distinct=37):
    """descriptors drawn from a small pool (+ a few flipped bits): many exact ties between train rows, so the lowest-index rule
    decides most matches"""
    rng = np.random.Generator(np.random.PCG64(seed))
    pool = rng.integers(0, 256, size=(distinct, 32), dtype=np.uint8)
    d = pool[rng.integers(0, distinct, size=n)].copy()
    flip = rng.integers(0, 4, size=n)
    for i in np.nonzero(flip == 0)[0]:
        d[i, rng.integers(0, 32)] ^= np.uint8(1 << rng.integers(0, 8))
    return d


def test_batch_device_matrix_core(gpu, oracle):
    """large batches run the matrix-core kernel (k_match_fp4: NQ = 2 below 32 jobs): ragged counts incl. 0 / 1 / a partial last chunk / the full
    stride, tie-heavy sets (lowest train index must win), every job against the oracle bit for bit"""
    from dvslam_amd import BFMatcher
    from dvslam_amd._lib import DeviceBuffer
    m = BFMatcher()
    P, S = 10, 2024
    nq = np.array([2024, 2000, 1, 0, 1999, 129, 128, 127, 2005, 33], np.int32)
    nt = np.array([2024, 1, 2000, 500, 0, 2005, 128, 129, 257, 1900], np.int32)
    Q = np.stack([_tie_heavy(S, 300 + p) if p % 2 else synth.make_descriptors(S, 300 + p) for p in range(P)])
    T = np.stack([_tie_heavy(S, 300 + p + (0 if p % 4 == 1 else 50)) if p % 2 else synth.make_descriptors(S, 400 + p) for p in range(P)])
    T[0, 1500] = Q[0, 7]; T[0, 1700] = Q[0, 7]; T[2, 1999] = Q[2, 0]
    dq = DeviceBuffer(Q.nbytes).upload(Q); dt = DeviceBuffer(T.nbytes).upload(T)
    dnq = DeviceBuffer(P * 4).upload(nq); dnt = DeviceBuffer(P * 4).upload(nt)
    di = DeviceBuffer(P * S * 4); dd = DeviceBuffer(P * S * 4)
    for rep in range(2):   # second call: same buffers, same results
        m.match_batch_device(dq.ptr, dnq.ptr, S, dt.ptr, dnt.ptr, S, P, di.ptr, dd.ptr)
        m.synchronize()
        idx = di.download(np.int32, P * S).reshape(P, S); d = dd.download(np.int32, P * S).reshape(P, S)
        for p in range(P):
            i2, d2 = oracle.match(Q[p, :nq[p]], T[p, :nt[p]])
            assert (idx[p, :nq[p]] == i2).all() and (d[p, :nq[p]] == d2).all(), (rep, p)


@pytest.mark.parametrize("env", [{"DVS_MATCH_MFMA": "0"}, {"DVS_MATCH_LDS": "0"}])
def test_match_switches_are_result_neutral(gpu, oracle, monkeypatch, env):
    """DVS_MATCH_MFMA=0 (popcount kernels for every job count) and DVS_MATCH_LDS=0 (few jobs by k_match<16, 1> instead of k_match_lds):
    a batch of large jobs and a few-jobs sequence against the oracle under either switch"""
    from dvslam_amd import BFMatcher
    from dvslam_amd._lib import DeviceBuffer
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    m = BFMatcher()     # (the switches are read when the handle is created / at the launch)
    for F, S in ((9, 2024), (3, 2024)):
        rng = np.random.Generator(np.random.PCG64(5 + F))
        D = rng.integers(0, 256, size=(F, S, 32), dtype=np.uint8)
        n = np.array([S - 7 * p for p in range(F)], np.int32)
        dd = DeviceBuffer(D.nbytes).upload(D); dn = DeviceBuffer(F * 4).upload(n)
        di = DeviceBuffer(F * S * 4); dq = DeviceBuffer(F * S * 4)
        m.match_sequence_device(dd.ptr, dn.ptr, S, F, 0, 0, di.ptr, dq.ptr)
        m.synchronize()
        idx = di.download(np.int32, F * S).reshape(F, S); dist = dq.download(np.int32, F * S).reshape(F, S)
        assert (idx[0, :n[0]] == -1).all()
        for p in range(1, F):
            i2, d2 = oracle.match(D[p, :n[p]], D[p - 1, :n[p - 1]])
            assert (idx[p, :n[p]] == i2).all() and (dist[p, :n[p]] == d2).all(), (env, F, p)


def test_matrix_core_extreme_popcounts_and_unaligned_bases(gpu, oracle):
    """rows of 0 and 256 set bits (the ninth k-step carries -64 |t| per train row: both ends of its range), rows that differ in one bit,
    and descriptor bases that are not 16-byte aligned (those jobs take the popcount kernel): all against the oracle"""
    from dvslam_amd import BFMatcher
    from dvslam_amd._lib import DeviceBuffer
    m = BFMatcher()
    P, S = 9, 2024
    rng = np.random.Generator(np.random.PCG64(77))
    Q = rng.integers(0, 256, size=(P, S, 32), dtype=np.uint8); T = rng.integers(0, 256, size=(P, S, 32), dtype=np.uint8)
    Q[0, ::3] = 0; Q[0, 1::3] = 255; T[0, ::5] = 0; T[0, 2::5] = 255
    T[1, :] = 255; T[1, 1000, 31] = 0x7F                      # all ones but one bit in one row
    T[2, :] = 0; T[2, 77, 0] = 1; Q[2, :] = 0
    Q[3, :] = 255; T[3, :] = 255                               # every distance 0: index 0 must win everywhere
    T[4] = Q[4][::-1]                                          # a permutation: every query has an exact twin
    nq = np.full(P, S, np.int32); nt = np.full(P, S, np.int32); nt[5] = 1; nq[6] = 31
    pad = 8
    dq = DeviceBuffer(Q.nbytes + pad); dt = DeviceBuffer(T.nbytes + pad)
    dnq = DeviceBuffer(P * 4).upload(nq); dnt = DeviceBuffer(P * 4).upload(nt)
    di = DeviceBuffer(P * S * 4); dd = DeviceBuffer(P * S * 4)
    for off in (0, pad):
        dq.upload(np.concatenate([np.zeros(off, np.uint8), Q.reshape(-1)])); dt.upload(np.concatenate([np.zeros(off, np.uint8), T.reshape(-1)]))
        m.match_batch_device(dq.ptr + off, dnq.ptr, S, dt.ptr + off, dnt.ptr, S, P, di.ptr, dd.ptr)
        m.synchronize()
        idx = di.download(np.int32, P * S).reshape(P, S); d = dd.download(np.int32, P * S).reshape(P, S)
        for p in range(P):
            i2, d2 = oracle.match(Q[p, :nq[p]], T[p, :nt[p]])
            assert (idx[p, :nq[p]] == i2).all() and (d[p, :nq[p]] == d2).all(), (off, p)
    assert (idx[3] == 0).all() and (d[3] == 0).all() and (d[4] == 0).all()


def test_matrix_core_full_machine_shape(gpu, oracle):
    """from 32 jobs on the matrix-core kernel takes four query tiles per wavefront (k_match_fp4<4>): 40 ragged jobs — counts around the
    512-query workgroup, the 128-row chunk and the 32-row tile boundaries, tie-heavy sets — against the oracle, and the XCD re-deal of
    the job index (40 = a multiple of 8)"""
    from dvslam_amd import BFMatcher
    from dvslam_amd._lib import DeviceBuffer
    m = BFMatcher()
    P, S = 40, 2024
    rng = np.random.Generator(np.random.PCG64(91))
    nq = rng.integers(1, S + 1, size=P).astype(np.int32); nt = rng.integers(1, S + 1, size=P).astype(np.int32)
    nq[:8] = [2024, 513, 512, 511, 129, 128, 33, 1]; nt[:8] = [2024, 1, 31, 32, 33, 127, 128, 129]
    Q = np.stack([_tie_heavy(S, 900 + p) if p % 3 == 0 else synth.make_descriptors(S, 900 + p) for p in range(P)])
    T = np.stack([_tie_heavy(S, 900 + p) if p % 3 == 0 else synth.make_descriptors(S, 1900 + p) for p in range(P)])
    dq = DeviceBuffer(Q.nbytes).upload(Q); dt = DeviceBuffer(T.nbytes).upload(T)
    dnq = DeviceBuffer(P * 4).upload(nq); dnt = DeviceBuffer(P * 4).upload(nt)
    di = DeviceBuffer(P * S * 4); dd = DeviceBuffer(P * S * 4)
    m.match_batch_device(dq.ptr, dnq.ptr, S, dt.ptr, dnt.ptr, S, P, di.ptr, dd.ptr)
    m.synchronize()
    idx = di.download(np.int32, P * S).reshape(P, S); d = dd.download(np.int32, P * S).reshape(P, S)
    for p in range(P):
        i2, d2 = oracle.match(Q[p, :nq[p]], T[p, :nt[p]])
        assert (idx[p, :nq[p]] == i2).all() and (d[p, :nq[p]] == d2).all(), p


@pytest.mark.parametrize("P,seed", [(9, 1), (16, 2), (33, 3), (48, 4)])
def test_matrix_core_randomized_densities(gpu, oracle, P, seed):
    """both shapes of the matrix-core kernel (NQ = 2 below 32 jobs, NQ = 4 from 32 on) on random jobs: counts anywhere in 1..2024, bit
    densities from 2 % to 98 % per set (|t| and |q| far from 128: the row constants and the f32 keys over their whole range), train rows
    that repeat (ties: the lowest index must win)"""
    from dvslam_amd import BFMatcher
    from dvslam_amd._lib import DeviceBuffer
    m = BFMatcher()
    S = 2024
    rng = np.random.Generator(np.random.PCG64(1000 + seed))
    nq = rng.integers(1, S + 1, size=P).astype(np.int32); nt = rng.integers(1, S + 1, size=P).astype(np.int32)
    Q = np.zeros((P, S, 32), np.uint8); T = np.zeros((P, S, 32), np.uint8)
    for p in range(P):
        dq, dt = rng.uniform(0.02, 0.98, size=2)
        Q[p] = np.packbits(rng.random((S, 256)) < dq, axis=1)
        T[p] = np.packbits(rng.random((S, 256)) < dt, axis=1)
        rep = rng.integers(0, max(int(nt[p]), 1), size=S // 4)          # a quarter of the train rows repeat earlier rows
        T[p, S // 2:S // 2 + len(rep)] = T[p, rep]
        hit = rng.integers(0, max(int(nt[p]), 1), size=8)               # and a few queries have exact twins
        Q[p, :8] = T[p, hit]
    dq_ = DeviceBuffer(Q.nbytes).upload(Q); dt_ = DeviceBuffer(T.nbytes).upload(T)
    dnq = DeviceBuffer(P * 4).upload(nq); dnt = DeviceBuffer(P * 4).upload(nt)
    di = DeviceBuffer(P * S * 4); dd = DeviceBuffer(P * S * 4)
    m.match_batch_device(dq_.ptr, dnq.ptr, S, dt_.ptr, dnt.ptr, S, P, di.ptr, dd.ptr)
    m.synchronize()
    idx = di.download(np.int32, P * S).reshape(P, S); d = dd.download(np.int32, P * S).reshape(P, S)
    for p in range(P):
        i2, d2 = oracle.match(Q[p, :nq[p]], T[p, :nt[p]])
        assert (idx[p, :nq[p]] == i2).all() and (d[p, :nq[p]] == d2).all(), (P, p, int(nq[p]), int(nt[p]))


def test_sequence_device_matrix_core(gpu, oracle):
    """the bench's shape on the MFMA kernel: frames of up to 2024 descriptors, frame p against p - 1 (every set used in
    both roles), frame 0 against a predecessor elsewhere or against nothing"""
    from dvslam_amd import BFMatcher
    from dvslam_amd._lib import DeviceBuffer
    m = BFMatcher()
    F, S = 9, 2024
    n = np.array([2005, 2024, 0, 64, 1999, 2000, 1, 1500, 2011], np.int32)
    D = np.stack([_tie_heavy(S, 700 + p, 101) if p in (4, 5) else synth.make_descriptors(S, 700 + p) for p in range(F)])
    D[1, 3] = D[0, 9]; D[1, 4] = D[0, 9]; D[0, 2004] = D[0, 9]
    prev = synth.make_descriptors(S, 799); nprev = np.array([1777], np.int32)
    dd = DeviceBuffer(D.nbytes).upload(D); dn = DeviceBuffer(F * 4).upload(n)
    dp = DeviceBuffer(prev.nbytes).upload(prev); dnp = DeviceBuffer(4).upload(nprev)
    di = DeviceBuffer(F * S * 4); dx = DeviceBuffer(F * S * 4)
    for with_prev in (True, False):
        m.match_sequence_device(dd.ptr, dn.ptr, S, F, dp.ptr if with_prev else 0, dnp.ptr if with_prev else 0, di.ptr, dx.ptr)
        m.synchronize()
        idx = di.download(np.int32, F * S).reshape(F, S); d = dx.download(np.int32, F * S).reshape(F, S)
        for p in range(F):
            t = (prev[:nprev[0]] if with_prev else prev[:0]) if p == 0 else D[p - 1, :n[p - 1]]
            i2, d2 = oracle.match(D[p, :n[p]], t)
            assert (idx[p, :n[p]] == i2).all() and (d[p, :n[p]] == d2).all(), (with_prev, p)
