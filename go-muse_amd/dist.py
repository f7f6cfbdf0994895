"""One process per GPU: the comparison Group is sharded by contiguous row
ranges; every (reference, series) pair is independent, so the only
exchange step is the gather of each shard's top-N candidate records
(top_n x 24 B per rank) -- one all_gather over RCCL/xGMI (backend "nccl") or
gloo (CPU tests) -- followed by the Results merge (muse_merge_records).

Process set-up (bench.py does exactly this): initialise torch's GPU runtime first
(torch.cuda.set_device(local_rank); dist.init_process_group("nccl", ...)), THEN create the
Engine -- torch bundles its own HIP runtime and must be the first to open the device.

Correctness condition: with grouped runs every label group must
live on ONE shard (shard on group boundaries); Run(nil) is always exact.
"""
import numpy as np

from . import binding as B
from .muse import merge_records


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
