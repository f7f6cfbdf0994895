"""Pins the CPU oracle against every known-answer table the reference's own
tests hold for the XCorr / Batch.Run path (tests/golden/reference_tables.json,
transcribed from xcorr_test.go, muse_batch_test.go, muse_test.go)."""
import numpy as np
import pytest


def test_next_pow2(golden, oracle):           # xcorr_test.go:20-38
    for c in golden["next_pow2"]:
        assert oracle.next_pow2(c["val"]) == c["expected"]
    # exact for powers of two up to 2^28; non powers round up
    for k in range(0, 29):
        assert oracle.next_pow2(2.0 ** k) == 2 ** k
    for v, e in [(480, 512), (12, 16), (5, 8), (16385, 32768), (4097, 8192)]:
        assert oracle.next_pow2(v) == e


def test_znormalize(golden, oracle):          # xcorr_test.go:40-61
    for ts in golden["znormalize"]["cases"]:
        z, zero = oracle.znormalize(ts)
        assert not zero
        assert abs(float(np.sum(z * z)) - (len(ts) - 1)) <= golden["znormalize"]["tol"]
    _, zero = oracle.znormalize([3.0] * 7)
    assert zero


def test_zero_pad(golden, oracle):            # xcorr_test.go:63-85
    for c in golden["zero_pad"]:
        out = oracle.zero_pad(c["x"], c["n"])
        assert out.tolist() == [float(v) for v in c["expected"]]


def _check_sign(mv, sign):
    if sign > 0:
        assert mv > 0
    elif sign < 0:
        assert mv < 0
    else:
        assert mv == 0


def test_xcorr(golden, oracle):               # xcorr_test.go:86-202
    tol = golden["xcorr"]["tol"]
    for c in golden["xcorr"]["cases"]:
        cc, lag, mv = oracle.xcorr(c["x"], c["y"], len(c["x"]), c["normalize"])
        if c["cc"] is None:
            assert cc is None
        else:
            assert np.max(np.abs(cc - np.array(c["cc"], dtype=float))) <= tol
        assert lag == c["idx"]
        _check_sign(mv, c["sign"])


def test_xcorr_with_x(golden, oracle):        # xcorr_test.go:204-286
    tol = golden["xcorr_with_x"]["tol"]
    for c in golden["xcorr_with_x"]["cases"]:
        n = len(c["x"])                        # the test uses n = len(X) = 5
        X, _ = oracle.ref_spectrum(c["x"], n=n)
        cc, lag, mv, _ = oracle.xcorr_with_x(X, c["y"], n)
        if c["cc"] is None:
            assert cc is None
        else:
            assert np.max(np.abs(cc - np.array(c["cc"], dtype=float))) <= tol
        assert lag == c["idx"]
        _check_sign(mv, c["sign"])


def _group_ids(comp, group_by):
    ids, gid = {}, []
    for s in comp:
        key = tuple((k, s["labels"][k]) for k in sorted(group_by) if k in s["labels"]) \
            if group_by else tuple(sorted(s["labels"].items()))
        gid.append(ids.setdefault(key, len(ids)))
    return np.array(gid, dtype=np.int32), len(ids)


def _run_batch(oracle, case, abs_scores=True, one_group_per_series=False):
    comp = case["comp"]
    r = case["results"]
    if not comp:
        return []
    rows = np.array([s["y"] for s in comp], dtype=np.float64)
    lag, mv, _ = oracle.batch_scores(case["ref"], rows)
    if one_group_per_series:
        gid, G = None, 0
    else:
        gid, G = _group_ids(comp, case.get("group_by"))
    idx, lg, sc, _ = oracle.results(lag, mv, gid, G, abs_scores, r["max_lag"],
                                    r["top_n"], r["threshold"], r["sign_filter"])
    return [(comp[i]["labels"], int(l), float(s)) for i, l, s in zip(idx, lg, sc)]


def _compare(got, case):                       # compareScores, muse_test.go:11-39
    exp = case["expected"]
    assert len(got) == len(exp)
    for (labels, lag, score), e in zip(got, exp):
        if "lag_in" in e:
            assert lag in e["lag_in"]
        else:
            assert lag == e["lag"]
        assert abs(score - e["score"]) <= case["score_tol"]
        assert labels == e["labels"]


def test_batch_run_simple(golden, oracle):    # muse_batch_test.go:9-44
    _compare(_run_batch(oracle, golden["batch_run_simple"]), golden["batch_run_simple"])


def test_batch_run_multidim(golden, oracle):  # muse_batch_test.go:46-82
    _compare(_run_batch(oracle, golden["batch_run_multidim"]), golden["batch_run_multidim"])


def test_muse_run_simple(golden, oracle):     # muse_test.go:41-73
    c = golden["muse_run_simple"]
    _compare(_run_batch(oracle, c, abs_scores=False, one_group_per_series=True), c)


def test_muse_run_sign_filter(golden, oracle):  # muse_test.go:75-104 (+ fresh NEG)
    for key in ("muse_run_sign_filter_pass1", "muse_run_sign_filter_neg_fresh"):
        c = golden[key]
        _compare(_run_batch(oracle, c, abs_scores=False, one_group_per_series=True), c)


def test_muse_run_no_input(golden, oracle):   # muse_test.go:122-142
    c = golden["muse_run_no_input"]
    assert _run_batch(oracle, c, abs_scores=False) == []
    idx, lg, sc, mean = oracle.results(np.zeros(0, np.int32), np.zeros(0), None, 0,
                                       False, 10, 20, 0.0, 0)
    assert len(idx) == 0 and np.isnan(mean)


def test_sign_filter_tie_is_exact(golden, oracle):
    """SURVEY section 4 trap: cc[13] == cc[14] == -65/6 /sigma... in exact
    arithmetic; the oracle must flag it through its gap output."""
    c = golden["muse_run_simple"]
    X, n = oracle.ref_spectrum(c["ref"])
    assert n == 16
    cc, lag, mv, gap = oracle.xcorr_with_x(X, c["comp"][3]["y"], n)
    assert lag in (-3, -2) and gap < 1e-12
    assert abs(abs(cc[13]) - abs(cc[14])) < 1e-14


@pytest.mark.parametrize("N", [5, 8, 12, 16, 100, 480, 512, 1000])
def test_oracle_exactness_vs_long_double_direct(oracle, N):
    """The oracle's own error: FFT path vs direct long-double correlation."""
    rng = np.random.default_rng(N)
    ref = rng.standard_normal(N)
    y = rng.standard_normal(N) + 0.5 * np.roll(ref, 3)
    X, n = oracle.ref_spectrum(ref)
    cc, lag, mv, _ = oracle.xcorr_with_x(X, y, n)
    direct = oracle.xcorr_direct_ld(ref, y, n)
    assert np.max(np.abs(cc - direct)) < 1e-13
    mi = int(np.argmax(np.abs(direct)))
    assert (mi - n if mi > n // 2 else mi) == lag


def test_oracle_vs_numpy_restatement(oracle):
    """Independent numpy restatement of xCorrWithX at the north-star shape."""
    rng = np.random.default_rng(7)
    N = n = 4096
    ref = rng.standard_normal(N)
    rows = rng.standard_normal((16, N))
    rows[3] = 2.5 * ref + 1.0
    rows[5] = -np.roll(ref, 7)   # y delayed by 7 => negative lag (xcorr.go:103)
    rows[6] = 3.0
    lag, mv, gap = oracle.batch_scores(ref, rows, nthreads=2)
    xz = (ref - ref.mean()) / ref.std(ddof=1) / (N - 1)
    Xf = np.fft.rfft(xz)
    for i in range(16):
        y = rows[i]
        if y.std() == 0:
            assert lag[i] == 0 and mv[i] == 0
            continue
        yz = (y - y.mean()) / y.std(ddof=1)
        cc = np.fft.irfft(np.conj(np.fft.rfft(yz)) * Xf, n)
        mi = int(np.argmax(np.abs(cc)))
        assert abs(cc[mi] - mv[i]) <= 1e-12
        assert (mi - n if mi > n // 2 else mi) == lag[i]
    assert lag[3] == 0 and abs(mv[3] - 1.0) < 1e-12
    assert lag[5] == -7 and abs(mv[5] + 1.0) < 1e-12


def test_results_heap_topn_and_order(oracle):
    """results.go:55-87: top-N by |score|, Fetch descending, filters."""
    mv = np.array([0.2, -0.9, 0.5, 0.7, 0.1, -0.3])
    lag = np.array([0, 1, -20, 2, 0, 3], dtype=np.int32)
    idx, lg, sc, mean = oracle.results(lag, mv, None, 0, True, 10, 3, 0.0, 0)
    assert idx.tolist() == [1, 3, 5] and sc.tolist() == [0.9, 0.7, 0.3]
    assert abs(mean - (0.9 + 0.7 + 0.3) / 3) < 1e-15
    # signed (Muse) + NEG filter; threshold
    idx, lg, sc, _ = oracle.results(lag, mv, None, 0, False, 10, 3, 0.25, -1)
    assert idx.tolist() == [1, 5] and sc.tolist() == [-0.9, -0.3]
    # group max keeps the first on ties and carries its lag
    gid = np.array([0, 0, 1, 1, 2, 2], dtype=np.int32)
    idx, lg, sc, _ = oracle.results(lag, np.array([0.5, 0.5, 0.1, 0.4, 2.0, -3.0]),
                                    gid, 3, True, 10, 5, 0.0, 0)
    assert idx.tolist() == [4, 0, 3] and sc.tolist() == [1.0, 0.5, 0.4]


def test_fast_cpu_port_matches_oracle(oracle):
    """oracle/muse_cpu_fast.c (the TIMED cpu_baseline of bench.py: radix-4 Stockham FFT, -O3 -march=native) against the
    checker on the same rows: scores to 1e-12 relative, lags equal outside oracle-flagged ties; sigma == 0 rows,
    zero-padded lengths, several threads."""
    import numpy as np
    rng = np.random.default_rng(17)
    for N in (8, 100, 480, 512, 1000, 4096, 5000):
        M = 37
        ref = rng.standard_normal(N)
        rows = rng.standard_normal((M, N))
        rows[::5] += rng.uniform(-3, 3, size=(len(rows[::5]), 1)) * np.roll(ref, 3)
        rows[7] = 4.25                                  # sigma == 0 -> (0, 0.0)
        olag, omv, gap = oracle.batch_scores(ref, rows)
        for nt in (1, 3):
            lag, mv = oracle.fast_batch_scores(ref, rows, nthreads=nt)
            assert mv[7] == 0.0 and lag[7] == 0
            np.testing.assert_allclose(mv, omv, rtol=1e-12, atol=1e-15)
            assert np.all((lag == olag) | (gap < 1e-12)), N
