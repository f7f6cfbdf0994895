"""One process per GPU: the comparison Group is sharded by contiguous row
ranges; every (reference, series) pair is independent, so the only
exchange step is the gather of each shard's top-N candidate records
(top_n x 24 B per rank) -- one all_gather over RCCL/xGMI (backend "nccl") or
gloo (CPU tests) -- followed by the Results merge (muse_merge_records).

Process set-up (bench.py does exactly this): initialise torch's GPU runtime first
(torch.cuda.set_device(local_rank); dist.init_process_group("nccl", ...)), THEN create the
Engine -- torch bundles its own HIP runtime and must be the first to open the device.

Label-grouped Runs (Batch.Run(["graph"]), muse_batch.go:99-130) take SURVEY 8e's second branch: the members of a label
group may sit on ANY ranks (graphs interleaved over the rows, a cut in the middle of a graph), so every rank reports, per
label group and unfiltered, its winner among its own members (muse_batch_run_groups: G x 25 B), the ranks exchange those
records, and the per-group maximum (muse_batch.go:87: strictly greater replaces, the earlier row wins ties, a NaN first
member poisons the group) is taken BEFORE Results.passed and the top-N heap:
  * up to EXACT_FEED_MAX_GROUPS label groups: one all_gather; every rank then holds all G winners and feeds them through
    Results.Update in group order -- the reference's own feed (muse_batch.go:124-128), so exactly tied scores come back in
    the reference's order, also into a Results that earlier Runs (other Batches: BASELINE configs[4]) have filled;
  * more label groups: an all_to_all hands rank r the records of the groups [r Gs, (r+1) Gs) from every rank (1/W of the
    bytes of the all_gather per link), rank r merges, filters and pre-selects its slice's top-N, and the slices' candidates
    (top_n x 24 B) are gathered and merged like the ungrouped Run's.
Run(nil) -- every series its own group -- needs neither: per-shard top-N candidates are exact for any cut (run_sharded).
"""
import numpy as np

from . import binding as B
from .muse import EXACT_FEED_MAX_GROUPS, Score, feed_group_winners, merge_group_records, merge_group_winners, merge_records


def shard_bounds(total_rows, world_size, rank, align=2):
    """contiguous row range [lo, hi) of `rank`; shard starts are multiples of
    `align` (the fused kernel packs two series per workgroup pass)."""
    per = -(-total_rows // world_size)
    per = -(-per // align) * align
    lo = min(rank * per, total_rows)
    hi = min(lo + per, total_rows)
    return lo, hi


def shard_bounds_grouped(group_id, world_size, rank):
    """Row range [lo, hi) of `rank` for a GROUPED run (config 5: Batch.Run(["graph"]) over 8 GPUs): rows must be
    ordered so that every label group is contiguous (group_id non-decreasing or at least run-length contiguous);
    cuts fall on group boundaries nearest to the even split, so no group straddles two shards and the per-shard
    group maxima are exact.  Returns (lo, hi); a rank may get an empty range when there are fewer groups than ranks."""
    gid = np.asarray(group_id)
    M = len(gid)
    if M == 0:
        return 0, 0
    starts = np.flatnonzero(np.concatenate(([True], gid[1:] != gid[:-1])))   # first row of every run
    if len(np.unique(gid[starts])) != len(starts):
        raise ValueError("rows of one label group are not contiguous: reorder the Group before sharding")
    bounds = np.concatenate((starts, [M]))
    cuts = [0]
    for r in range(1, world_size):
        target = r * M / world_size
        c = int(bounds[np.argmin(np.abs(bounds - target))])
        cuts.append(max(c, cuts[-1]))
    cuts.append(M)
    return cuts[rank], cuts[rank + 1]


def gather_records(local_records, top_n, group=None, device=None):
    """all_gather of fixed-size (top_n records + count) buffers.  Returns the
    concatenated valid records of all ranks (on every rank)."""
    import torch
    import torch.distributed as dist

    cap = max(int(top_n), 1)
    local_records = np.ascontiguousarray(local_records, dtype=B.RECORD_DTYPE)[:cap]
    buf = np.zeros(cap + 1, dtype=B.RECORD_DTYPE)          # slot 0 carries the count
    buf[0]["series"] = len(local_records)
    buf[1:1 + len(local_records)] = local_records
    if not (dist.is_available() and dist.is_initialized()):
        return local_records.copy()
    world = dist.get_world_size(group)
    t = torch.from_numpy(buf.view(np.uint8).copy())
    if device is not None:
        t = t.to(device)
    out = torch.empty(world * t.numel(), dtype=torch.uint8, device=t.device)
    dist.all_gather_into_tensor(out, t, group=group)
    allb = out.cpu().numpy().view(B.RECORD_DTYPE).reshape(world, cap + 1)
    parts = [allb[r, 1:1 + int(allb[r, 0]["series"])] for r in range(world)]
    return np.concatenate(parts) if parts else np.zeros(0, dtype=B.RECORD_DTYPE)


def run_sharded(dbatch, series_offset, group_id=None, G=0, max_lag=10, top_n=20, threshold=0.0,
                sign_filter=0, abs_scores=True, group=None, device=None):
    """Batch.Run over a sharded Group: local fused pass + local top-N on this
    rank's GPU, one gather, merge.  group_id holds GLOBAL group ids of the local
    rows.  Returns (series, lag, score, mean_abs) with global series indices,
    identical on every rank."""
    rec = dbatch.run_shard(group_id, G, series_offset, max_lag, top_n, threshold, sign_filter, abs_scores)
    allrec = gather_records(rec, top_n, group=group, device=device)
    return merge_records(allrec, top_n)


# ------------------------------------------------------------------ label groups that straddle ranks
_STATE_DTYPE = np.uint8
_REC_BYTES = B.RECORD_DTYPE.itemsize


def _pack_groups(rec, state, G):
    """one rank's G records + G states as ONE byte buffer (a single collective per Run)"""
    rec = np.ascontiguousarray(rec, dtype=B.RECORD_DTYPE)
    state = np.ascontiguousarray(state, dtype=_STATE_DTYPE)
    if len(rec) != G or len(state) != G:
        raise ValueError("run_groups returned %d records / %d states for %d label groups" % (len(rec), len(state), G))
    buf = np.empty(G * (_REC_BYTES + 1), dtype=np.uint8)
    buf[:G * _REC_BYTES] = rec.view(np.uint8).reshape(-1)
    buf[G * _REC_BYTES:] = state
    return buf


def _unpack_groups(buf, W, G):
    buf = np.asarray(buf, dtype=np.uint8).reshape(W, G * (_REC_BYTES + 1))
    rec = np.ascontiguousarray(buf[:, :G * _REC_BYTES]).view(B.RECORD_DTYPE).reshape(W, G)
    state = np.ascontiguousarray(buf[:, G * _REC_BYTES:])
    return rec, state


def gather_group_records(rec, state, group=None, device=None):
    """all_gather of every rank's per-group records: -> ((W, G) records, (W, G) states), ranks in ascending row order
    (rank r holds the r-th row range), identical on every rank"""
    import torch
    import torch.distributed as dist

    G = len(rec)
    buf = _pack_groups(rec, state, G)
    if not (dist.is_available() and dist.is_initialized()):
        return _unpack_groups(buf, 1, G)
    world = dist.get_world_size(group)
    t = torch.from_numpy(buf)
    if device is not None:
        t = t.to(device)
    out = torch.empty(world * t.numel(), dtype=torch.uint8, device=t.device)
    dist.all_gather_into_tensor(out, t, group=group)
    return _unpack_groups(out.cpu().numpy(), world, G)


def exchange_group_slices(rec, state, group=None, device=None):
    """all_to_all of the per-group records: rank r receives, from every rank, the records of the label groups
    [r Gs, (r + 1) Gs), Gs = ceil(G / W) (groups beyond G: state 0).  -> ((W, Gs) records, (W, Gs) states, first group id)"""
    import torch
    import torch.distributed as dist

    G = len(rec)
    if not (dist.is_available() and dist.is_initialized()):
        r, s = _unpack_groups(_pack_groups(rec, state, G), 1, G)
        return r, s, 0
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    Gs = max(-(-G // world), 1)
    prec = np.zeros(world * Gs, dtype=B.RECORD_DTYPE)
    prec["series"] = -1
    pst = np.zeros(world * Gs, dtype=_STATE_DTYPE)
    prec[:G], pst[:G] = rec, state
    send = np.concatenate([_pack_groups(prec[r * Gs:(r + 1) * Gs], pst[r * Gs:(r + 1) * Gs], Gs) for r in range(world)])
    t = torch.from_numpy(send)
    if device is not None:
        t = t.to(device)
    out = torch.empty_like(t)
    dist.all_to_all_single(out, t, group=group)
    r, s = _unpack_groups(out.cpu().numpy(), world, Gs)
    return r, s, rank * Gs


def run_groups_sharded(dbatch, series_offset, group_id, G, abs_scores=True, group=None, device=None):
    """The per-group part of a label-grouped Run over a Group sharded by rows, label groups on ANY ranks: this rank's
    muse_batch_run_groups, one all_gather, muse_merge_group_winners.  group_id: GLOBAL group ids in [0, G) of the local
    rows.  -> (winners[G], state[G]), identical on every rank: state 1 = winners[g] is label group g's Score (global
    series index, clamped score, lag), 0 = the group has no member, 2 = its first member scores NaN (never passes)."""
    rec, state = dbatch.run_groups(group_id, G, series_offset, abs_scores=abs_scores)
    allrec, allst = gather_group_records(rec, state, group=group, device=device)
    return merge_group_winners(allrec, allst)


feed_results = feed_group_winners     # Batch.Run's ordered drain, one Score per label group (muse.py)


def run_grouped_sharded(dbatch, series_offset, group_id, G, max_lag=10, top_n=20, threshold=0.0, sign_filter=0,
                        abs_scores=True, group=None, device=None, exact_feed_max_groups=None, with_groups=False):
    """Batch.Run(groupByLabels) + Fetch over a Group sharded by rows, one process per GPU, label groups anywhere
    (muse_batch.go:99-130, results.go:46-87).  -> (series, lag, score, mean_abs) in Fetch order with GLOBAL series indices,
    identical on every rank; with_groups: the label group of every entry as a fifth item."""
    limit = EXACT_FEED_MAX_GROUPS if exact_feed_max_groups is None else exact_feed_max_groups
    rec, state = dbatch.run_groups(group_id, G, series_offset, abs_scores=abs_scores)
    if G <= limit:
        # every rank gets all G winners; muse_merge_group_records = per-group maximum, Results.passed, then the heap fed in
        # group order: the reference's feed into a fresh Results
        allrec, allst = gather_group_records(rec, state, group=group, device=device)
        out = merge_group_records(allrec, allst, max_lag, top_n, threshold, sign_filter)
        if not with_groups:
            return out
        win, _ = merge_group_winners(allrec, allst)
        gof = dict(zip(win["series"].tolist(), win["group"].tolist()))
        return out + (np.array([gof[int(i)] for i in out[0]], dtype=np.int64),)
    # very many label groups: rank r owns the groups of slice r -- merge, filter and pre-select there, gather top_n candidates
    srec, sst, g0 = exchange_group_slices(rec, state, group=group, device=device)
    win, wst = merge_group_winners(srec, sst)
    live = (wst == 1) & (np.abs(win["lag"].astype(np.int64)) <= max_lag) & (np.abs(win["score"]) >= threshold)
    if sign_filter > 0:
        live &= win["score"] > 0
    elif sign_filter < 0:
        live &= win["score"] < 0
    cand = win[live].copy()
    cand["group"] = np.nonzero(live)[0] + g0
    keep = np.lexsort((cand["group"], -np.abs(cand["score"])))[:max(int(top_n), 0)]   # the slice's best top_n, lower group id on ties
    allrec = gather_records(cand[np.sort(keep)], top_n, group=group, device=device)
    out = merge_records(allrec, top_n)
    if not with_groups:
        return out
    gof = dict(zip(allrec["series"].tolist(), allrec["group"].tolist()))
    return out + (np.array([gof[int(i)] for i in out[0]], dtype=np.int64),)


class ShardedBatch:
    """One rank's part of a Batch whose Comparison group is sharded by rows over the ranks of a process group
    (one process per GPU; bench.py --gpus N).  The mirror of Batch for that layout: `Results` is the caller's (it may be
    shared with other batches, results.go:55-72) and after Run it holds the same Scores on every rank.
      dbatch        DeviceBatch over this rank's rows [series_offset, series_offset + M_local)
      labels_of     (global series index, label group id) -> Labels of the Score (label bookkeeping is the caller's)"""

    def __init__(self, dbatch, series_offset, results, labels_of, group=None, device=None):
        self.dbatch, self.series_offset, self.Results, self.labels_of = dbatch, int(series_offset), results, labels_of
        self.group, self.device = group, device

    def Run(self, group_id, G):
        """group_id: GLOBAL label-group ids in [0, G) of the local rows (what indexLabelValues numbers, group.go:76-104)"""
        r = self.Results
        if G <= EXACT_FEED_MAX_GROUPS:
            win, st = run_groups_sharded(self.dbatch, self.series_offset, group_id, G, True, self.group, self.device)
            feed_results(r, win, st, self.labels_of)
            return None
        idx, lag, score, _, grp = run_grouped_sharded(self.dbatch, self.series_offset, group_id, G, r.MaxLag, r.TopN, r.Threshold,
                                                      r.SignFilter, True, self.group, self.device, with_groups=True)
        # the pre-selected candidates, fed in group order (muse_batch.go:124-128); among EXACTLY tied scores the survivor at the
        # TopN boundary may differ from a full feed (docs/HISTORY.md 8.3)
        for k in np.lexsort((idx, grp)):
            r.Update(Score(self.labels_of(int(idx[k]), int(grp[k])), int(lag[k]), float(score[k])))
        return None
