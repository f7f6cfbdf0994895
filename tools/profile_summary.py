#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output (kernel trace stats + PMC passes) into a small text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(sub, pat):
    return sorted(glob.glob(os.path.join(out, sub, "**", pat), recursive=True))


print("# rocprofv3 summary for", out)
for f in find("trace", "*kernel_stats.csv"):
    print("\n## kernel stats (%s)" % os.path.relpath(f, out))
    rows = list(csv.DictReader(open(f)))
    for r in rows[:12]:
        print("  %-60s calls=%s total_ns=%s avg_ns=%s pct=%s" % (
            r.get("Name", "")[:60], r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"), r.get("Percentage")))
for sub in ("pmc_fetch", "pmc_write", "pmc_sq1", "pmc_sq2"):
    for f in find(sub, "*counter_collection.csv"):
        acc = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r.get("Kernel_Name", "")][r.get("Counter_Name", "")].append(float(r.get("Counter_Value", 0)))
        print("\n## %s (%s)" % (sub, os.path.relpath(f, out)))
        for k, cs in acc.items():
            if "xcorr_fused" not in k and "xcorr_screen" not in k:
                continue
            for c, vals in sorted(cs.items()):
                print("  %-40s %-28s n=%d mean=%.6g" % (k[:40], c, len(vals), sum(vals) / len(vals)))

# HBM traffic per launch of the fused kernel, as MI355X_MICROARCH.md (HBM) prescribes for gfx950:
# FETCH_SIZE (KB) reads exactly half of a wide coalesced stream -> x2; WRITE_SIZE (KB) as is.
import json


def _mean(sub, counter):
    """mean over launches of the DOMINANT fused kernel (the NaN-pair fallback launch that follows the
    default kernel moves a few KB and must not be averaged in)"""
    per = defaultdict(list)
    for f in find(sub, "*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name", "")
            if ("xcorr_fused" in name or "xcorr_screen" in name) and r.get("Counter_Name") == counter:
                per[name].append(float(r.get("Counter_Value", 0)))
    if not per:
        return None, None
    name, best = max(per.items(), key=lambda kv: sum(kv[1]) / len(kv[1]))
    return sum(best) / len(best), name


(fetch, kname), (write, _) = _mean("pmc_fetch", "FETCH_SIZE"), _mean("pmc_write", "WRITE_SIZE")
if fetch is not None and write is not None:
    traffic = (2.0 * fetch + write) * 1024.0
    print("\n## HBM traffic per fused launch: 2*FETCH_SIZE + WRITE_SIZE = %.4g B (FETCH_SIZE %.4g KB, WRITE_SIZE %.4g KB)"
          % (traffic, fetch, write))
    json.dump({"rows": int(os.environ.get("PROFILE_ROWS", "1000000")), "length": 4096,
               "kernel": kname.split("(")[0].replace("void ", "").replace("muse::", "").strip(),
               "hbm_bytes_per_launch": traffic, "fetch_size_kb": fetch, "write_size_kb": write,
               "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE doubled (gfx950)"},
              open(os.path.join(out, "traffic.json"), "w"))
