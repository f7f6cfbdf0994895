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
