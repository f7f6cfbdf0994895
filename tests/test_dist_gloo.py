"""world_size-2 gloo test (CPU) of the sharded Batch.Run path:
each rank owns a contiguous row shard, produces its local top-N candidate
records, one all_gather of top_n x 24-byte records, then muse_merge_records.
On CPU the per-shard records come from the oracle (the GPU kernels are covered
by -m gpu tests); what is exercised here is exactly the N > 1 code in
go-muse_amd/dist.py plus the host-side merge in libmuse_hip.so."""
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


def _shard_records(muse, oracle_py, ref, rows, lo, hi, gid, G, args):
    """local Results.Update/Fetch over rows [lo, hi) -> records with GLOBAL indices"""
    lag, mv, _ = oracle_py.batch_scores(ref, rows[lo:hi])
    g_local = None if gid is None else gid[lo:hi]
    idx, lg, sc, _ = oracle_py.results(lag, mv, g_local, G, args["abs"], args["max_lag"], args["top_n"],
                                       args["thr"], args["sign"])
    rec = np.zeros(len(idx), dtype=muse.binding.RECORD_DTYPE)
    rec["series"] = idx + lo
    rec["lag"] = lg
    rec["score"] = sc
    rec["group"] = (idx + lo) if gid is None else gid[lo:hi][idx]
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
        out = {}
        for name, grouped in (("ungrouped", False), ("grouped", True)):
            args = dict(abs=True, max_lag=8, top_n=12, thr=0.1, sign=0)
            if grouped:   # label groups of 7 consecutive series; shards cut on group boundaries
                gid = (np.arange(M) // 7).astype(np.int32)
                G = int(gid.max()) + 1
                lo, hi = muse.dist.shard_bounds(M, world, rank, align=14)
            else:
                gid, G = None, 0
                lo, hi = muse.dist.shard_bounds(M, world, rank)
            rec = _shard_records(muse, oracle_py, ref, rows, lo, hi, gid, G, args)
            allrec = muse.dist.gather_records(rec, args["top_n"])
            s, l, sc, mean = muse.merge_records(allrec, args["top_n"])
            lag, mv, _ = oracle_py.batch_scores(ref, rows)
            es, el, esc, emean = oracle_py.results(lag, mv, gid, G, True, 8, 12, 0.1, 0)
            out[name] = (s.tolist() == es.tolist() and l.tolist() == el.tolist()
                         and sc.tolist() == esc.tolist() and mean == emean, len(allrec), (lo, hi))
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_sharded_run_world_size_2_gloo():
    muse = pkg()
    muse.build.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, out in res:
        for name, (ok, nrec, bounds) in out.items():
            assert ok, (rank, name, bounds)
            assert nrec <= 24


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
