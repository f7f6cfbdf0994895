"""GPU parity tests for series longer than 65 536 samples (FFT lengths 2^17 ... 2^20, xcorr_huge.hip): the reference has no
length limit (xcorr.go:19-24, 160-197, muse_batch.go:33-37).  Everything goes through the C ABI and is compared with the CPU
oracle on byte-identical inputs: scores within 1e-6 relative, lags exact (except rounding-decided ties, asserted absent)."""
import numpy as np
import pytest

from _load import pkg

pytestmark = pytest.mark.gpu

SCORE_RTOL = 1e-6
SCORE_ATOL = 1e-12
TIE_GAP = 1e-12


@pytest.fixture(scope="module")
def muse():
    m = pkg()
    m.build.build()
    import torch
    if torch.cuda.is_available():
        torch.cuda.init()
    return m


@pytest.fixture(scope="module")
def eng(muse):
    return muse.get_engine(0)


def _check(lag, mv, olag, omv, gap):
    lag, mv, olag, omv = map(np.asarray, (lag, mv, olag, omv))
    nan_o = np.isnan(omv)
    assert np.array_equal(np.isnan(mv), nan_o), (mv, omv)
    ok = ~nan_o
    err = np.abs(mv[ok] - omv[ok])
    assert np.all(err <= SCORE_RTOL * np.abs(omv[ok]) + SCORE_ATOL), (float(err.max()), mv, omv)
    tie = (gap < TIE_GAP) & ok
    assert not ((lag != olag) & ~tie & ok).any(), (lag, olag)
    assert np.all(lag[nan_o] == 0)
    return float(np.max(err / np.maximum(np.abs(omv[ok]), 1e-300)))


def _rows(rng, M, N, ref):
    rows = rng.standard_normal((M, N))
    for i in range(0, M, 3):
        rows[i] += rng.uniform(-4, 4) * np.roll(ref, int(rng.integers(-N // 3, N // 3)))
    rows[1] = 2.5                                                      # sigma == 0: (nil, 0, 0) -> score 0, lag 0
    rows[2] *= 1e50                                                    # scales far apart inside a pair (2 | 3)
    rows[3] *= 1e-120
    if M > 6:
        rows[5, N // 7] = np.nan                                       # a NaN series beside a healthy partner (4 | 5)
        rows[6, N - 1] = np.inf
    if M > 8:
        rows[8] = rows[0]                                              # exact copy: identical score and lag
    return rows


@pytest.mark.parametrize("N,M", [(65537, 11), (100000, 13), (131072, 12), (262144, 9), (300000, 7), (524288, 5), (1000003, 5), (1048576, 4)])
def test_huge_all_scores_match_oracle(muse, eng, oracle, N, M):
    """muse_batch_scores at N = n and zero-padded lengths up to 2^20 (odd row counts: the last series has no partner), with
    constant, NaN, Inf, 1e50- and 1e-120-scaled rows inside pairs; the reference spectrum against the oracle's."""
    rng = np.random.default_rng(7000 + N % 9973)
    ref = rng.standard_normal(N)
    ref[N // 2:N // 2 + 40] += 6.0
    rows = _rows(rng, M, N, ref)
    dg = muse.DeviceGroup.from_rows(eng, rows)
    db = muse.DeviceBatch(eng, dg, ref)
    assert db.n == oracle.next_pow2(N) and db.n > 65536
    X, n = oracle.ref_spectrum(ref)
    X = X[0::2] + 1j * X[1::2]
    Xd = db.spectrum()
    scale = np.max(np.abs(X))
    assert np.max(np.abs(Xd - X)) <= 1e-11 * scale
    lag, mv = db.scores()
    olag, omv, gap = oracle.batch_scores(ref, rows, nthreads=4)
    worst = _check(lag, mv, olag, omv, gap)
    assert worst < 1e-9
    assert mv[1] == 0.0 and lag[1] == 0
    if M > 8:
        assert mv[8] == mv[0] and lag[8] == lag[0]
    lag2, mv2 = db.scores()                                            # bit-reproducible
    assert np.array_equal(lag, lag2) and np.array_equal(mv, mv2, equal_nan=True)
    db.close()
    dg.close()


def test_huge_run_with_label_groups_and_filters(muse, eng, oracle):
    """Batch.Run over a Group of 100 000-sample series with label groups (muse_batch_run / _run_groups) against the oracle's
    post-processing of the oracle's scores; Muse.Run (muse_batch_run_rows) on the same rows."""
    rng = np.random.default_rng(7101)
    N, M, G = 100000, 26, 7
    ref = rng.standard_normal(N)
    rows = rng.standard_normal((M, N))
    for i in range(M):
        rows[i] += rng.uniform(-1.5, 1.5) * np.roll(ref, int(rng.integers(-12, 13)))
    rows[4] = -1.0
    gid = rng.integers(0, G, M).astype(np.int32)
    dg = muse.DeviceGroup.from_rows(eng, rows)
    db = muse.DeviceBatch(eng, dg, ref)
    olag, omv, gap = oracle.batch_scores(ref, rows, nthreads=4)
    for kw in (dict(max_lag=10, top_n=4, threshold=0.0, sign_filter=0), dict(max_lag=50000, top_n=20, threshold=0.05, sign_filter=0)):
        got = db.run(gid, G, abs_scores=True, **kw)
        exp = oracle.results(olag, omv, gid, G, True, kw["max_lag"], kw["top_n"], kw["threshold"], kw["sign_filter"])
        assert list(got[0]) == list(exp[0]) and list(got[1]) == list(exp[1])
        assert np.allclose(got[2], exp[2], rtol=SCORE_RTOL, atol=SCORE_ATOL)
    got = db.run(None, 0, max_lag=15, top_n=5, threshold=0.0, sign_filter=1, abs_scores=False)
    exp = oracle.results(olag, omv, None, 0, False, 15, 5, 0.0, 1)
    assert list(got[0]) == list(exp[0]) and np.allclose(got[2], exp[2], rtol=SCORE_RTOL, atol=SCORE_ATOL)
    probe = muse.DeviceGroup(eng, N, 0)
    tmpl = muse.DeviceBatch.like(db, probe)
    sub = rows[:5]
    win, st = tmpl.run_rows(sub, abs_scores=False)                      # Muse.Run: one label group from host memory
    sc = np.clip(omv[:5], -1.0, 1.0)
    best = int(np.argmax(np.abs(sc)))
    assert st == 1 and int(win["series"]) == best and int(win["lag"]) == int(olag[best])
    assert abs(win["score"] - sc[best]) <= SCORE_RTOL * abs(sc[best]) + SCORE_ATOL
    for h in (tmpl, probe, db, dg):
        h.close()


@pytest.mark.parametrize("Nx,Ny,n", [(70000, 70000, 131072), (100000, 131072, 0), (131072, 90001, 262144), (262144, 262144, 0)])
def test_huge_two_sided_matches_oracle(muse, eng, oracle, Nx, Ny, n):
    """xCorr (xcorr.go:102-153) for pairs of series longer than 65 536 samples, normalised and raw, full cc: against the
    oracle's xcorr; a constant x or y gives (nil, 0, 0) when normalised; a NaN sample gives NaN."""
    rng = np.random.default_rng(7200 + Nx % 1000 + Ny % 77)
    M = 5
    x = rng.standard_normal((M, Nx))
    y = rng.standard_normal((M, Ny))
    L = min(Nx, Ny)
    y[0, Ny - L:] += 2.0 * np.roll(x[0, Nx - L:], 37)
    y[1, :] = 4.0                                                       # sigma(y) == 0
    x[2, :] = -0.5                                                      # sigma(x) == 0
    y[3, Ny // 2] = np.nan
    gx, gy = muse.DeviceGroup.from_rows(eng, x), muse.DeviceGroup.from_rows(eng, y)
    nn = max(n, Nx, Ny)
    for normalize in (True, False):
        cc, lag, mv, nil = muse.xcorr_groups(gx, gy, n, normalize, want_cc=True)
        for i in range(M):
            occ, olag, omv = oracle.xcorr(x[i], y[i], n, normalize)
            if occ is None:
                assert nil[i] == 1 and lag[i] == 0 and mv[i] == 0.0
                continue
            assert nil[i] == 0
            if np.isnan(omv):
                assert np.isnan(mv[i]) and lag[i] == 0 and np.isnan(cc[i]).all()
                continue
            assert len(occ) == nn
            ref_scale = max(np.max(np.abs(occ)), 1e-300)
            assert np.max(np.abs(cc[i] - occ)) <= 1e-9 * ref_scale, (i, normalize)
            assert abs(mv[i] - omv) <= SCORE_RTOL * abs(omv) + SCORE_ATOL
            srt = np.sort(np.abs(occ))
            if srt[-1] - srt[-2] > 1e-9 * srt[-1]:
                assert lag[i] == olag
    gx.close()
    gy.close()


def test_huge_single_pair_entry_points(muse, eng, oracle):
    """muse_xcorr_with_x / muse_xcorr with the full cc slice at n = 131072 (xcorr_test.go-style access)."""
    rng = np.random.default_rng(7300)
    N = 90000
    ref = rng.standard_normal(N)
    y = rng.standard_normal(N) + 1.7 * np.roll(ref, -21)
    cc, lag, mv = eng.xcorr_with_x(ref, y)
    X, n = oracle.ref_spectrum(ref)
    occ, olag, omv, gap = oracle.xcorr_with_x(X, y, n)
    assert n == 131072 and lag == olag and abs(mv - omv) <= SCORE_RTOL * abs(omv)
    assert np.max(np.abs(cc - occ)) <= 1e-9 * np.max(np.abs(occ))
    cc2, lag2, mv2 = eng.xcorr(ref, y, 131072, True)
    occ2, olag2, omv2 = oracle.xcorr(ref, y, 131072, True)
    assert lag2 == olag2 and abs(mv2 - omv2) <= SCORE_RTOL * abs(omv2)
    assert np.max(np.abs(cc2 - occ2)) <= 1e-9 * np.max(np.abs(occ2))
    cc3, lag3, mv3 = eng.xcorr_with_x(ref, np.full(N, 3.0))             # sigma(y) == 0 -> (nil, 0, 0)
    assert cc3 is None and lag3 == 0 and mv3 == 0.0


def _numpy_xcorr(x, y, n, normalize):
    """xcorr.go:102-153 restated with numpy's FFT (any n): the checker for lengths the C oracle's naive DFT cannot reach in time"""
    n = max(n, len(x), len(y))

    def zn(v):
        v = v - v.sum() / len(v)
        sd = np.sqrt(np.sum((v - v.mean()) ** 2) / (len(v) - 1))
        return None if sd == 0 else v / sd
    if normalize:
        x, y = zn(x), zn(y)
        if x is None or y is None:
            return None, 0, 0.0
    xp, yp = np.zeros(n), np.zeros(n)
    xp[n - len(x):] = x
    yp[n - len(y):] = y
    cc = np.fft.irfft(np.fft.rfft(xp) * np.conj(np.fft.rfft(yp)), n) * n      # gonum's Sequence is unnormalised
    cc *= 1.0 / (float(n) * (n - 1)) if normalize else 1.0 / n
    mi = 0
    best = 0.0
    a = np.abs(cc)
    mi = int(np.argmax(a)) if a.max() > 0 else 0                                # (argmax returns the first maximum)
    mv = cc[mi]
    return cc, (mi - n if mi > n // 2 else mi), mv


@pytest.mark.parametrize("lenx,leny,n", [(10000, 10000, 0), (9000, 12001, 0), (20000, 5000, 30000), (70001, 70001, 0), (300000, 250000, 0)])
def test_xcorr_at_any_n(muse, eng, lenx, leny, n):
    """xCorr takes the n it is given (xcorr.go:104-106) and gonum transforms any length: for n that is not a power of two the
    circular correlation is folded out of a power-of-two one at L >= 2 n (muse_xcorr: single pair, fold and argmax on the host) --
    n = 10 000 ... 300 000 against a numpy restatement, normalised and raw, full cc; a constant series is nil."""
    rng = np.random.default_rng(lenx + leny)
    x = rng.standard_normal(lenx) * 3.0 + 1.0
    y = rng.standard_normal(leny)
    L = min(lenx, leny)
    y[leny - L:] += 1.5 * np.roll(x[lenx - L:], 11)
    for normalize in (True, False):
        cc, lag, mv = eng.xcorr(x, y, n, normalize)
        occ, olag, omv = _numpy_xcorr(x, y, n, normalize)
        assert len(cc) == len(occ) == max(n, lenx, leny)
        assert np.max(np.abs(cc - occ)) <= 1e-9 * np.max(np.abs(occ))
        assert lag == olag and abs(mv - omv) <= SCORE_RTOL * abs(omv)
    cc, lag, mv = eng.xcorr(np.full(lenx, 2.0), y, n, True)
    assert cc is None and lag == 0 and mv == 0.0
    with pytest.raises(muse.MuseError) as ei:
        eng.xcorr(np.zeros(600000), np.zeros(600000), 0, False)                 # beyond 2^19 only powers of two are built
    assert ei.value.status == muse.binding.MUSE_ERR_UNSUPPORTED


def test_lengths_above_the_limit_are_refused(muse, eng):
    N = (1 << 20) + 1
    dg = muse.DeviceGroup(eng, N, 0)
    with pytest.raises(muse.MuseError) as ei:
        muse.DeviceBatch(eng, dg, np.arange(N, dtype=np.float64))
    assert ei.value.status == muse.binding.MUSE_ERR_UNSUPPORTED
    dg.close()
    ref = np.sin(np.arange(70000) * 0.01)
    dg = muse.DeviceGroup(eng, 70000, 0)
    with pytest.raises(muse.MuseError) as ei:
        muse.DeviceBatch(eng, dg, np.full(70000, 2.0))                  # muse_batch.go:39-41
    assert ei.value.status == muse.binding.MUSE_ERR_ZERO_STD
    db = muse.DeviceBatch(eng, dg, ref)                                  # an empty group scores nothing
    lag, mv = db.scores()
    assert len(lag) == 0
    db.close()
    dg.close()
