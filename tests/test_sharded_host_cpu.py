"""CPU test of the IN-PROCESS sharded Batch.Run (SURVEY 8e inside one process: one context + host thread per device, DESIGN
section 7a).  The per-shard device batches are replaced by stubs with the same run_shard / run_groups contracts whose scores
come from the CPU checker; everything around them -- the cut into row ranges, the straddle test, the host threads, the merges
in libmuse_hip.so (muse_merge_records / muse_merge_group_records), the ordered drain into Results -- is the product code the
GPUs run under.  The outcome must be what the checker's Results gives over ALL rows at once, also for label groups that
straddle every shard and for a group whose first member scores NaN.  (The GPU kernels behind run_shard / run_groups:
tests/test_gpu_parity.py::test_batch_run_sharded_over_engine_list and host/muse_host_test.cpp.)"""
import math

import numpy as np
import pytest

from _load import pkg


def _clamp(mv, abs_scores):
    return np.minimum(np.abs(mv), 1.0) if abs_scores else np.clip(mv, -1.0, 1.0)


class StubShardBatch:
    """DeviceBatch stand-in for rows [lo, hi): scores precomputed by the checker"""

    def __init__(self, muse, oracle, lag, mv, dgroup):
        self.muse, self.oracle, self.lag, self.mv, self.dgroup = muse, oracle, lag, mv, dgroup

    def run_shard(self, group_id=None, G=0, series_offset=0, max_lag=10, top_n=20, threshold=0.0, sign_filter=0, abs_scores=True):
        dt = self.muse.binding.RECORD_DTYPE
        uniq, gl = np.unique(np.asarray(group_id), return_inverse=True)
        idx, lg, sc, _ = self.oracle.results(self.lag, self.mv, gl.astype(np.int32), len(uniq), abs_scores, max_lag, top_n, threshold, sign_filter)
        rec = np.zeros(len(idx), dtype=dt)
        rec["series"], rec["lag"], rec["score"], rec["group"] = idx + series_offset, lg, sc, np.asarray(group_id)[idx]
        return rec

    def run_groups(self, group_id, G, series_offset=0, abs_scores=True):
        """muse_batch_run_groups' contract: per label group the winner among the members whose score is a number (series -1 if
        none), unfiltered, and the state 0 no member / 1 first member a number / 2 first member NaN"""
        dt = self.muse.binding.RECORD_DTYPE
        rec = np.zeros(G, dtype=dt)
        rec["series"] = -1
        rec["group"] = np.arange(G)
        state = np.zeros(G, dtype=np.uint8)
        s = _clamp(self.mv, abs_scores)
        for i, g in enumerate(np.asarray(group_id)):
            if state[g] == 0:
                state[g] = 2 if math.isnan(s[i]) else 1
            if math.isnan(s[i]):
                continue
            if rec[g]["series"] < 0 or abs(s[i]) > abs(rec[g]["score"]):     # strictly greater replaces: the first wins ties
                rec[g] = (i + series_offset, s[i], self.lag[i], g)
        return rec, state


@pytest.mark.parametrize("exact_feed", [True, False])
@pytest.mark.parametrize("n_shards", [2, 3, 5])
def test_in_process_sharded_run_equals_results_over_all_rows(monkeypatch, n_shards, exact_feed):
    muse = pkg()
    from oracle import oracle_py as oracle
    rng = np.random.default_rng(17 + n_shards)
    M, N = 241, 64
    ref = rng.standard_normal(N)
    rows = rng.standard_normal((M, N))
    rows[::6] += 1.5 * np.roll(ref, 3)
    rows[5, 7] = np.nan                       # series 5 = the FIRST member of graph 5: the group's score is NaN
    rows[5 + 24 * 3, 2] = np.nan              # a later member of graph 5 ... of another shard
    rows[9] = 0.75                            # sigma == 0: score 0
    graphs, hosts = np.arange(M) % 24, np.arange(M) // 24
    lag, mv, _ = oracle.batch_scores(ref, rows)
    series = [muse.NewSeries(rows[i], muse.NewLabels({"graph": "g%d" % graphs[i], "host": "h%d" % hosts[i]})) for i in range(M)]
    refs = muse.NewSeries(ref, muse.NewLabels({"graph": "ref"}))

    class FakeEngine:
        pass
    engines = [FakeEngine() for _ in range(n_shards)]

    def fake_shards(self, engs):               # Group._device_shards without device memory: the same cut (dist.shard_bounds)
        out = []
        for r, e in enumerate(engs):
            lo, hi = muse.dist.shard_bounds(len(self.registry), len(engs), r)
            out.append((e, ("rows", lo, hi), lo, hi))
        return out
    monkeypatch.setattr(muse.Group, "_device_shards", fake_shards)
    # exact_feed: one Score per label group through Results.Update (what a Run over up to 65 536 groups does); otherwise the
    # pre-selecting paths a Run over more groups takes (per-shard top-N / per-group records + muse_merge_*)
    monkeypatch.setattr(muse.muse, "EXACT_FEED_MAX_GROUPS", 65536 if exact_feed else 0)
    monkeypatch.setattr(muse.muse, "DeviceBatch",
                        lambda e, dg, r: StubShardBatch(muse, oracle, lag[dg[1]:dg[2]], mv[dg[1]:dg[2]], dg))

    for by, gid, max_lag, top, thr, sf in [(None, np.arange(M), N, 20, 0.0, 0), (["graph"], graphs, N, 20, 0.0, 0),
                                           (["host"], hosts, N, 4, 0.0, 0), (["graph"], graphs, 5, 8, 0.05, 1),
                                           (["graph", "host"], np.arange(M), N, 300, 0.0, 0)]:
        g = muse.NewGroup("targets")
        g.Add(*series)
        b = muse.Batch.__new__(muse.Batch)     # (the constructor probes the reference on a device)
        b.Concurrency, b.Comparison, b.Results = 1, g, muse.NewResults(max_lag, top, thr, sf)
        b._engines, b._engine, b._ref, b._db, b._shard_db = engines, engines[0], ref, None, None
        b.Run(by)
        got, mean = b.Results.Fetch()
        # group ids as indexLabelValues numbers them: first appearance in insertion order
        first = {}
        dense = np.array([first.setdefault(int(x), len(first)) for x in gid], dtype=np.int32)
        oi, ol, osc, omean = oracle.results(lag, mv, dense, len(first), True, max_lag, top, thr, sf)
        assert [s.Lag for s in got] == list(ol)
        assert [s.PercentScore for s in got] == list(osc)
        assert [s.Labels.ID() for s in got] == [series[int(i)].Labels().ID() for i in oi]
        assert mean == omean or (math.isnan(mean) and math.isnan(omean))
        assert all(s.Labels.Get("graph")[0] != "g5" for s in got) or by != ["graph"]   # NaN-first group never passes


def test_feed_group_winners_skips_only_no_op_updates():
    """feed_group_winners never constructs a Score that Results.Update would ignore (fails passed(), or not strictly above a full
    heap's rising minimum): the heap it leaves must be the plain loop's -- order among exact ties included -- for fresh and
    pre-filled Results, heavy ties, every filter."""
    muse = pkg()
    rng = np.random.default_rng(5)
    for trial in range(40):
        G = int(rng.integers(1, 30000 if trial % 8 == 0 else 400))
        win = np.zeros(G, dtype=muse.binding.RECORD_DTYPE)
        win["series"] = rng.permutation(G)
        win["score"] = rng.choice([1.0, 0.5, 0.25, -0.5, 0.0], size=G) if trial % 2 else np.round(rng.uniform(-1, 1, G), 2)
        win["lag"] = rng.integers(-20, 21, size=G)
        win["group"] = np.arange(G)
        state = rng.choice([0, 1, 1, 1, 2], size=G).astype(np.uint8)
        top, max_lag, thr, sf = int(rng.integers(0, 25)), int(rng.integers(0, 25)), float(rng.choice([0.0, 0.3])), int(rng.integers(-1, 2))
        a, b = muse.NewResults(max_lag, top, thr, sf), muse.NewResults(max_lag, top, thr, sf)
        for k in range(int(rng.integers(0, 30))):              # history from an earlier Run
            a.Update(muse.Score(muse.NewLabels({"pre": str(k)}), 0, [0.5, 1.0, 0.75][k % 3]))
        b.scores = [muse.Score(s.Labels, s.Lag, s.PercentScore) for s in a.scores]   # identical starting heaps, slot for slot
        lab = lambda i, g: muse.NewLabels({"row": str(i), "g": str(g)})
        muse.muse.feed_group_winners(a, win, state, lab)
        for g in range(G):                                     # the plain feed: every group's Score through Update
            if state[g] == 1:
                b.Update(muse.Score(lab(int(win["series"][g]), g), int(win["lag"][g]), float(win["score"][g])))
        ga, ma = a.Fetch()
        gb, mb = b.Fetch()
        assert [(s.Labels.labels, s.Lag, s.PercentScore) for s in ga] == [(s.Labels.labels, s.Lag, s.PercentScore) for s in gb]
        assert ma == mb or (math.isnan(ma) and math.isnan(mb))
