"""GPU tests of the one-launch reduction of small Runs (reduce_kernels.hip, small_groups_kernel: up to 32 768 series in up to 2 048
label groups; run with -m gpu on an MI355X).  Through the C ABI, against (a) a numpy restatement of Batch.scoreSingle + Results
(muse_batch.go:74-89, results.go:46-86) fed with the device's own scores, and (b) the general four-launch path, reached for the
same rows by padding the label-group count past 2 048 with empty groups (an empty group never reaches the heap: results.go:56-59)."""
import numpy as np
import pytest

from _load import pkg

pytestmark = pytest.mark.gpu


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


def _clamp(mv, abs_scores):
    v = np.abs(mv) if abs_scores else mv.copy()
    return np.clip(v, -1.0, 1.0)  # (NaN stays NaN)


def _restate_run(mv, lag, gid, G, max_lag, top_n, threshold, sign_filter, abs_scores):
    """one Score per label group in group order (first member, replaced by a strictly greater |score|: muse_batch.go:78-89), the
    Results filter and the N largest |score| (results.go:46-72), returned in Fetch order"""
    v = _clamp(mv, abs_scores)
    recs = []
    for g in range(G):
        idx = np.flatnonzero(gid == g) if gid is not None else np.array([g])
        if idx.size == 0:
            continue
        w = idx[0]
        if not np.isnan(v[w]):
            for i in idx[1:]:
                if abs(v[i]) > abs(v[w]):
                    w = i
        s, lg = v[w], lag[w]
        ok = abs(lg) <= max_lag and abs(s) >= threshold and (sign_filter == 0 or (s > 0 and sign_filter == 1) or (s < 0 and sign_filter == -1))
        if ok:
            recs.append((w, lg, s))
    # min-heap of size top_n on |score|: a later record replaces the root only when strictly greater (results.go:60-71)
    import heapq
    heap = []
    for k, (w, lg, s) in enumerate(recs):
        if len(heap) < top_n:
            heapq.heappush(heap, (abs(s), k))
        elif abs(s) > heap[0][0]:
            heapq.heapreplace(heap, (abs(s), k))
    kept = sorted(k for _, k in heap)
    return {recs[k][0] for k in kept}, {recs[k][0]: recs[k] for k in kept}


def _rows(rng, M, N, ref):
    rows = rng.standard_normal((M, N))
    for i in range(0, M, 3):
        rows[i] += rng.uniform(-4, 4) * np.roll(ref, int(rng.integers(-12, 13)))
    return rows


@pytest.mark.parametrize("M,N,G", [(6, 8, 3), (5000, 8, 100), (5000, 480, 100), (32768, 8, 2048), (700, 30, 700), (1, 16, 1)])
def test_small_run_equals_the_restatement_and_the_general_path(muse, eng, M, N, G):
    rng = np.random.default_rng(M * 31 + N)
    ref = rng.standard_normal(N)
    rows = _rows(rng, M, N, ref)
    if M > 40:
        rows[5] = 2.5               # constant: NaN score
        rows[17, 1] = np.nan
        rows[M // 2] = rows[M // 2 - 1]  # an exact tie, put into one label group below: the first wins (muse_batch.go:87)
    gid = rng.integers(0, G, M).astype(np.int32)
    if M > 40:
        gid[M // 2] = gid[M // 2 - 1]
    if G > 4:
        gid[gid == 3] = 2           # an empty label group
        gid[5] = 4                  # a group whose FIRST member scores NaN ...
        gid[:5][gid[:5] == 4] = 0
    dg = muse.DeviceGroup.from_rows(eng, rows)
    db = muse.DeviceBatch(eng, dg, ref)
    lag, mv = db.scores()
    for kw in (dict(max_lag=10, top_n=20, threshold=0.0, sign_filter=0, abs_scores=True),
               dict(max_lag=3, top_n=7, threshold=0.4, sign_filter=0, abs_scores=True),
               dict(max_lag=N, top_n=50, threshold=0.1, sign_filter=-1, abs_scores=False),
               dict(max_lag=N, top_n=3000, threshold=0.0, sign_filter=1, abs_scores=False)):
        s, l, v, mean = db.run(gid, G, **kw)
        want_set, want = _restate_run(mv, lag, gid, G, **kw)
        assert set(s.tolist()) == want_set, (kw, sorted(set(s.tolist()) ^ want_set)[:8])
        for si, li, vi in zip(s, l, v):
            assert (li, vi) == (want[si][1], want[si][2])
        assert np.all(np.abs(v[:-1]) >= np.abs(v[1:]))                       # Fetch order: descending |score|
        if len(v):
            assert abs(mean - np.mean(np.abs(v))) <= 1e-12
        else:
            assert np.isnan(mean)
        # the general path on the same rows: label groups padded past 2 048 with empty ones
        s2, l2, v2, mean2 = db.run(gid, G + 2048, **kw)
        assert s.tolist() == s2.tolist() and l.tolist() == l2.tolist() and v.tolist() == v2.tolist()
        assert mean == mean2 or (np.isnan(mean) and np.isnan(mean2))
    for abs_scores in (True, False):
        rec, st = db.run_groups(gid, G, 7, abs_scores=abs_scores)
        rec2, st2 = db.run_groups(gid, G + 2048, 7, abs_scores=abs_scores)
        assert rec.tobytes() == rec2[:G].tobytes() and st.tolist() == st2[:G].tolist() and not st2[G:].any()
        v = _clamp(mv, abs_scores)
        for g in range(G):
            idx = np.flatnonzero(gid == g)
            if idx.size == 0:
                assert st[g] == 0 and rec[g]["series"] == -1
                continue
            assert st[g] == (2 if np.isnan(v[idx[0]]) else 1)
            num = idx[~np.isnan(v[idx])]
            if num.size == 0:
                assert rec[g]["series"] == -1
                continue
            w = num[np.argmax(np.abs(v[num]))]      # (argmax: the first of equal maxima)
            assert (rec[g]["series"], rec[g]["lag"], rec[g]["score"], rec[g]["group"]) == (w + 7, lag[w], v[w], g)
    db.close()
    dg.close()


@pytest.mark.parametrize("M", [1500, 10000, 32768])
def test_small_run_ungrouped_and_repeated(muse, eng, M):
    """no label map: every series its own group (muse_batch.go:60-66 with an empty label set; up to 32 768 series go through one
    launch of one thread per series); the same batch run again and again (the stamp advances per Run), interleaved with a Run on a
    second batch of the same context and with grouped Runs on the same batch (the smaller record buffer, then the larger again)"""
    rng = np.random.default_rng(77 + M)
    N = 60
    ref = rng.standard_normal(N)
    rows = _rows(rng, M, N, ref)
    rows[9] = -1.0
    dg = muse.DeviceGroup.from_rows(eng, rows)
    db = muse.DeviceBatch(eng, dg, ref)
    other = muse.DeviceBatch(eng, dg, np.roll(ref, 5))
    lag, mv = db.scores()
    kw = dict(max_lag=20, top_n=25, threshold=0.2, sign_filter=0, abs_scores=True)
    want_set, want = _restate_run(mv, lag, None, M, **kw)
    first = None
    gid = (np.arange(M) % 50).astype(np.int32)
    grouped = db.run(gid, 50, **kw)
    for it in range(50 if M < 5000 else 12):
        s, l, v, mean = db.run(None, 0, **kw)
        if it % 7 == 0:
            other.run(None, 0, **kw)
        if it % 5 == 0:
            again = db.run(gid, 50, **kw)
            assert all(a.tolist() == b.tolist() for a, b in zip(again[:3], grouped[:3]))
        assert set(s.tolist()) == want_set
        if first is None:
            first = (s.tolist(), l.tolist(), v.tolist())
        assert (s.tolist(), l.tolist(), v.tolist()) == first
    for si, li, vi in zip(*first):
        assert (li, vi) == (want[si][1], want[si][2])
    other.close()
    db.close()
    dg.close()
    # a batch made after the close takes over the context's pinned record buffer: same answer
    dg = muse.DeviceGroup.from_rows(eng, rows)
    db = muse.DeviceBatch(eng, dg, ref)
    s, l, v, mean = db.run(None, 0, **kw)
    assert (s.tolist(), l.tolist(), v.tolist()) == first
    db.close()
    dg.close()


@pytest.mark.parametrize("M,N", [(150000, 32), (70001, 100)])
def test_direct_topn_equals_the_general_path(muse, eng, M, N):
    """Run(nil) over more than 65 536 series: the chunks' candidates are selected and written into pinned slots by ONE launch
    (reduce_kernels.hip, topn_ungrouped_kernel).  Against the general path (group_final + topn + copies), reached for the same rows
    through the identity label map: the same series, lags and scores in the same order, with exact ties planted across and inside
    chunks, for top_n up to the device limit, with filters that leave chunks empty, and with a series offset (a shard's Run)."""
    rng = np.random.default_rng(M)
    ref = rng.standard_normal(N)
    rows = rng.standard_normal((M, N))
    rows[::5] += rng.uniform(-3, 3, (len(rows[::5]), 1)) * np.roll(ref, 2)[None, :]
    for i in rng.integers(1, M, 400):            # exact copies of another row: tied scores, far apart and adjacent
        rows[i] = rows[int(rng.integers(0, i))]
    rows[4097] = rows[4096] = rows[4095]
    rows[10] = 1.25                              # NaN score
    rows[8000:8192] = ref * 2.0                  # 192 series that all clamp to 1.0 inside one chunk and across a chunk boundary
    dg = muse.DeviceGroup.from_rows(eng, rows)
    db = muse.DeviceBatch(eng, dg, ref)
    ident = np.arange(M, dtype=np.int32)
    for kw in (dict(max_lag=N, top_n=20, threshold=0.0, sign_filter=0, abs_scores=True),
               dict(max_lag=N, top_n=256, threshold=0.0, sign_filter=0, abs_scores=True),
               dict(max_lag=2, top_n=1, threshold=0.5, sign_filter=0, abs_scores=True),
               dict(max_lag=N, top_n=50, threshold=0.9999, sign_filter=1, abs_scores=False),
               dict(max_lag=0, top_n=7, threshold=2.0, sign_filter=0, abs_scores=True)):      # nothing passes
        a = db.run(None, 0, **kw)
        b = db.run(ident, M, **kw)
        assert a[0].tolist() == b[0].tolist() and a[1].tolist() == b[1].tolist() and a[2].tolist() == b[2].tolist(), kw
        assert a[3] == b[3] or (np.isnan(a[3]) and np.isnan(b[3]))
        ra = db.run_shard(None, 0, series_offset=12345, **kw)
        rb = db.run_shard(ident, M, series_offset=12345, **kw)
        assert ra["series"].tolist() == rb["series"].tolist() and ra["score"].tolist() == rb["score"].tolist()
        assert ra["lag"].tolist() == rb["lag"].tolist()
        assert ra["series"].tolist() == (a[0] + 12345).tolist()
    assert len(db.run(None, 0, max_lag=N, top_n=256, threshold=0.0, sign_filter=0, abs_scores=True)[0]) == 256
    db.close()
    dg.close()
