"""world_size-2 gloo test (CPU) of the sharded Batch.Run path:
each rank owns a contiguous row shard, produces its local top-N candidate
records, one all_gather of top_n x 24-byte records, then muse_merge_records.
On CPU each rank's DeviceBatch is replaced by a stub with the same run_shard
signature whose records come from the oracle (the GPU kernels are covered by
-m gpu tests, incl. run_sharded over backend nccl at world size 1); everything
after it -- dist.run_sharded, the all_gather, the merge in libmuse_hip.so --
is the code the ranks run on GPUs."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from _load import ROOT, pkg


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class StubBatch:
    """Stands in for DeviceBatch on a box without a GPU: same run_shard signature and contract (this shard's top-N
    candidate records with GLOBAL series indices and global group ids), computed by the CPU checker.  Everything
    downstream of it -- dist.run_sharded -> gather_records (all_gather) -> merge_records (libmuse_hip.so) -- is the
    product code path the ranks run on GPUs."""

    def __init__(self, muse, oracle_py, ref, local_rows):
        self.muse, self.oracle_py, self.ref, self.rows = muse, oracle_py, ref, local_rows
        self.calls = 0

    def run_shard(self, group_id=None, G=0, series_offset=0, max_lag=10, top_n=20, threshold=0.0, sign_filter=0,
                  abs_scores=True):
        self.calls += 1
        rec_dtype = self.muse.binding.RECORD_DTYPE
        if len(self.rows) == 0:
            return np.zeros(0, dtype=rec_dtype)
        lag, mv, _ = self.oracle_py.batch_scores(self.ref, self.rows)
        if group_id is None:
            gl, Gl = None, 0
        else:   # the checker wants dense local ids; the records carry the global ones
            uniq, gl = np.unique(np.asarray(group_id), return_inverse=True)
            gl, Gl = gl.astype(np.int32), len(uniq)
        idx, lg, sc, _ = self.oracle_py.results(lag, mv, gl, Gl, abs_scores, max_lag, top_n, threshold, sign_filter)
        rec = np.zeros(len(idx), dtype=rec_dtype)
        rec["series"] = idx + series_offset
        rec["lag"] = lg
        rec["score"] = sc
        rec["group"] = (idx + series_offset) if group_id is None else np.asarray(group_id)[idx]
        return rec


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        muse = pkg()
        from oracle import oracle_py
        rng = np.random.default_rng(42)              # same data on every rank
        M, N = 1001, 64
        ref = rng.standard_normal(N)
        rows = rng.standard_normal((M, N))
        rows[::9] += np.roll(ref, 3) * rng.uniform(0.5, 3.0, (len(rows[::9]), 1))
        rows[500] = 2.0                               # sigma == 0
        lag, mv, _ = oracle_py.batch_scores(ref, rows)
        out = {}
        cases = [("ungrouped", None), ("grouped by 7", (np.arange(M) // 7).astype(np.int32)),
                 # two label groups only: with three ranks one shard is EMPTY and must still take part in the gather
                 ("two groups", (np.arange(M) >= 400).astype(np.int32))]
        for name, gid in cases:
            kw = dict(max_lag=8, top_n=12, threshold=0.1, sign_filter=0, abs_scores=True)
            if gid is None:
                G = 0
                lo, hi = muse.dist.shard_bounds(M, world, rank)
            else:
                G = int(gid.max()) + 1
                lo, hi = muse.dist.shard_bounds_grouped(gid, world, rank)
            stub = StubBatch(muse, oracle_py, ref, rows[lo:hi])
            s, l, sc, mean = muse.dist.run_sharded(stub, lo, None if gid is None else gid[lo:hi], G, **kw)
            es, el, esc, emean = oracle_py.results(lag, mv, gid, G, True, 8, 12, 0.1, 0)
            out[name] = (stub.calls == 1 and s.tolist() == es.tolist() and l.tolist() == el.tolist()
                         and sc.tolist() == esc.tolist() and mean == emean, (lo, hi))
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def _run_world(world):
    muse = pkg()
    muse.build.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_sharded_run_world_size_2_gloo():
    for rank, out in _run_world(2):
        for name, (ok, bounds) in out.items():
            assert ok, (rank, name, bounds)


def test_sharded_run_world_size_3_with_an_empty_shard_gloo():
    res = _run_world(3)
    empties = 0
    for rank, out in res:
        for name, (ok, bounds) in out.items():
            assert ok, (rank, name, bounds)
        empties += int(out["two groups"][1][0] == out["two groups"][1][1])
    assert empties >= 1       # the case under test did occur


def test_sharded_run_world_size_8_gloo():
    """eight ranks -- the size the driver's node runs -- with six EMPTY shards in the two-group case (every rank still takes part
    in the gather and ends with the same records)"""
    res = _run_world(8)
    empties = 0
    for rank, out in res:
        for name, (ok, bounds) in out.items():
            assert ok, (rank, name, bounds)
        empties += int(out["two groups"][1][0] == out["two groups"][1][1])
    assert empties == 6


def test_shard_bounds_cover_and_align():
    muse = pkg()
    for total in (0, 1, 7, 1000, 1_000_001):
        for world in (1, 2, 3, 8):
            spans = [muse.dist.shard_bounds(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and (a % 2 == 0 or a == total)


def test_shard_bounds_grouped_cuts_on_group_boundaries():
    import importlib
    dist_mod = importlib.import_module("go-muse_amd").dist
    rng = np.random.default_rng(0)
    sizes = rng.integers(1, 40, size=57)
    gid = np.repeat(np.arange(len(sizes)), sizes)
    M = len(gid)
    for world in (1, 2, 3, 8, 100):
        prev = 0
        owners = {}
        for rank in range(world):
            lo, hi = dist_mod.shard_bounds_grouped(gid, world, rank)
            assert lo == prev and lo <= hi <= M
            prev = hi
            for g in np.unique(gid[lo:hi]):
                assert g not in owners, "group straddles two shards"
                owners[g] = rank
        assert prev == M and len(owners) == len(sizes)
        if world <= 8:
            per = [dist_mod.shard_bounds_grouped(gid, world, r) for r in range(world)]
            assert max(h - l for l, h in per) <= M / world + 40      # near-even split
    with pytest.raises(ValueError):
        dist_mod.shard_bounds_grouped(np.array([0, 1, 0]), 2, 0)


# ---------------------------------------------------------------- label groups that straddle ranks (second branch)
def _clamp(mv, abs_scores):
    return np.minimum(np.abs(mv), 1.0) if abs_scores else np.clip(mv, -1.0, 1.0)


class StubGroupsBatch:
    """DeviceBatch stand-in with muse_batch_run_groups' contract (per label group, unfiltered: this shard's winner among the
    members whose score is a number + the group's state on this shard), scores from the CPU checker."""

    def __init__(self, muse, lag, mv):
        self.muse, self.lag, self.mv, self.calls = muse, lag, mv, 0

    def run_groups(self, group_id, G, series_offset=0, abs_scores=True):
        import math
        self.calls += 1
        rec = np.zeros(G, dtype=self.muse.binding.RECORD_DTYPE)
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


def _grouped_workload():
    """two (ref, Group) pairs of different lengths (BASELINE configs[4] in small): graphs INTERLEAVED over the rows so that
    every label group has members on every rank, planted bit-identical series in different graphs (exact ties at the top),
    a graph whose first member scores NaN, a constant series"""
    out = []
    for k, (M, N, graphs) in enumerate(((603, 64, 41), (410, 48, 29))):
        rng = np.random.default_rng(77 + k)
        ref = rng.standard_normal(N)
        rows = rng.standard_normal((M, N))
        rows[::5] += 1.3 * np.roll(ref, 2)
        strong = 1.7 * ref + 0.05 * rng.standard_normal(N)
        for i in (3, 100, 101, 222, 345, 346, 347, 401):       # bit-identical series in eight different graphs (i % graphs differ)
            rows[i] = strong
        rows[7, 5] = np.nan                                     # series 7 is the FIRST member of graph 7: the group scores NaN
        rows[7 + 3 * graphs, 1] = np.nan                        # a later NaN member of the same graph (another rank)
        rows[11] = 0.25                                         # sigma == 0
        gid = (np.arange(M) % graphs).astype(np.int32)          # interleaved: graph g = rows g, g + graphs, g + 2 graphs, ...
        out.append((ref, rows, gid, graphs))
    return out


def _grouped_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        muse = pkg()
        from oracle import oracle_py
        D = muse.dist
        work = _grouped_workload()
        scores = [oracle_py.batch_scores(ref, rows)[:2] for ref, rows, _, _ in work]
        ok = {}
        # (a) one Batch, fresh Results, four filter settings, both exchange paths (the all_gather + exact feed, and the
        #     all_to_all slices a Run over more than EXACT_FEED_MAX_GROUPS label groups takes -- forced by limit 0)
        ref, rows, gid, G = work[0]
        lag, mv = scores[0]
        lo, hi = D.shard_bounds(len(rows), world, rank)
        for max_lag, top, thr, sf in ((64, 6, 0.0, 0), (64, 50, 0.0, 0), (4, 5, 0.05, 1), (64, 3, 0.2, -1)):
            want = oracle_py.results(lag, mv, gid, G, True, max_lag, top, thr, sf)
            for limit in (None, 0):
                stub = StubGroupsBatch(muse, lag[lo:hi], mv[lo:hi])
                got = D.run_grouped_sharded(stub, lo, gid[lo:hi], G, max_lag, top, thr, sf, True, exact_feed_max_groups=limit,
                                            with_groups=True)
                same = (got[1].tolist() == want[1].tolist() and got[2].tolist() == want[2].tolist()
                        and (got[3] == want[3] or (np.isnan(got[3]) and np.isnan(want[3])))
                        and got[4].tolist() == gid[got[0]].tolist() and stub.calls == 1)
                if limit is None:      # the exact feed: the very series the reference's feed keeps, tie for tie
                    same = same and got[0].tolist() == want[0].tolist()
                else:                  # pre-selected slices: equal scores may swap; the multiset of (score, lag) is the reference's
                    same = same and sorted(zip(np.abs(mv[got[0]]).clip(max=1).tolist(), lag[got[0]].tolist())) == \
                        sorted(zip(want[2].tolist(), want[1].tolist()))
                ok["run %s limit %s" % ((max_lag, top, thr, sf), limit)] = bool(same)
        # (b) configs[4]'s flow: two Batches of different lengths Run(["graph"]) into ONE shared Results through ShardedBatch,
        #     one Fetch: what the reference's feed gives over the union (batch by batch, group by group)
        shared = muse.NewResults(64, 7, 0.0, muse.SignFilter_ANY)
        all_lag, all_mv, all_gid, base = [], [], [], 0
        for k, ((ref, rows, gid, G), (lag, mv)) in enumerate(zip(work, scores)):
            lo, hi = D.shard_bounds(len(rows), world, rank)
            sb = D.ShardedBatch(StubGroupsBatch(muse, lag[lo:hi], mv[lo:hi]), lo, shared,
                                lambda i, g, k=k: muse.NewLabels({"batch": str(k), "row": str(i), "graph": "g%d" % g}))
            sb.Run(gid[lo:hi], G)
            all_lag.append(lag)
            all_mv.append(mv)
            all_gid.append(gid + base)
            base += G
        got, mean = shared.Fetch()
        oi, ol, osc, omean = oracle_py.results(np.concatenate(all_lag), np.concatenate(all_mv), np.concatenate(all_gid), base,
                                               True, 64, 7, 0.0, 0)
        M0 = len(work[0][1])
        expect = [("0" if i < M0 else "1", str(i if i < M0 else i - M0), int(l), float(v)) for i, l, v in zip(oi, ol, osc)]
        have = [(s.Labels.labels["batch"], s.Labels.labels["row"], s.Lag, s.PercentScore) for s in got]
        ok["shared Results"] = have == expect and mean == omean
        ok["ties at the top"] = len({h[3] for h in have[:5]}) == 1           # (the planted copies really tie)
        # (c) fewer rows than ranks can hold in pairs: at world size 8 the last rank's shard is EMPTY and still takes part in both
        #     exchange paths (the all_gather of G records and the all_to_all slices)
        rng = np.random.default_rng(5)
        tref, trows = rng.standard_normal(16), rng.standard_normal((13, 16))
        trows[4] = trows[9]
        tgid = (np.arange(13) % 5).astype(np.int32)
        tlag, tmv = oracle_py.batch_scores(tref, trows)[:2]
        lo, hi = D.shard_bounds(13, world, rank)
        want = oracle_py.results(tlag, tmv, tgid, 5, True, 16, 4, 0.0, 0)
        for limit in (None, 0):
            stub = StubGroupsBatch(muse, tlag[lo:hi], tmv[lo:hi])
            got = D.run_grouped_sharded(stub, lo, tgid[lo:hi], 5, 16, 4, 0.0, 0, True, exact_feed_max_groups=limit, with_groups=True)
            ok["tiny rows limit %s" % limit] = bool(got[1].tolist() == want[1].tolist() and got[2].tolist() == want[2].tolist()
                                                    and (limit is not None or got[0].tolist() == want[0].tolist()))
        ok["_empty"] = lo == hi
        q.put((rank, ok, [h[:2] for h in have]))
    finally:
        dist.destroy_process_group()


def _run_grouped_world(world):
    muse = pkg()
    muse.build.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grouped_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize("world", [2, 3, 8])
def test_grouped_run_with_straddling_label_groups_gloo(world):
    """Batch.Run(["graph"]) over a Group sharded by rows with one process per GPU, graphs interleaved over all ranks, exact
    ties planted: every rank ends with the reference's Results over ALL rows (oracle.results = muse_batch.go:56-93,
    results.go:46-87 restated), through both exchange paths and through a Results shared by two Batches (configs[4])."""
    res = _run_grouped_world(world)
    firsts = None
    assert any(ok["_empty"] for _, ok, _ in res) == (world == 8)     # (at eight ranks the tiny case leaves one rank without rows)
    for rank, ok, have in res:
        for name, good in ok.items():
            assert good or name == "_empty", (rank, name)
        firsts = firsts or have
        assert have == firsts                                       # every rank holds the same Results
