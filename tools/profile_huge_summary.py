#!/usr/bin/env python3
"""Summary of tools/profile_huge.sh for one length: per kernel the launches, mean duration and measured HBM bytes per launch
(2 x FETCH_SIZE + WRITE_SIZE, KB units -> bytes), and per all-scores pass the sum against the algorithmic 8 N + 16 bytes per series."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

out, N, GB = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])


def short(name):
    name = re.sub(r"^void\s+", "", name.strip()).replace("(anonymous namespace)::", "").replace("muse::", "")
    return re.sub(r"\(.*$", "", name)


dur, calls = defaultdict(float), defaultdict(int)
for f in glob.glob(os.path.join(out, "trace_%d" % N, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
        calls[k] += 1
ctr = {"FETCH_SIZE": defaultdict(float), "WRITE_SIZE": defaultdict(float)}
cnt = {"FETCH_SIZE": defaultdict(int), "WRITE_SIZE": defaultdict(int)}
for sub, name in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    for f in glob.glob(os.path.join(out, "%s_%d" % (sub, N), "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                k = short(r["Kernel_Name"])
                ctr[name][k] += float(r["Counter_Value"])
                cnt[name][k] += 1
M = max(8, int(GB * (1 << 30) / (8 * N)) // 2 * 2)
passes = 9  # huge_bench.py: 3 warm-up + 6 timed all-scores passes per length
print("## N = %d (group of %d series, %.1f GB): %d all-scores passes traced" % (N, M, M * 8.0 * N / 1e9, passes))
tot_us, tot_b = 0.0, 0.0
for k in sorted(dur, key=lambda k: -dur[k]):
    if not (k.startswith("huge_") or k.startswith("xcorr_")):
        continue
    us = dur[k] / calls[k]
    fb = ctr["FETCH_SIZE"][k] / max(cnt["FETCH_SIZE"][k], 1) * 1024.0 * 2.0
    wb = ctr["WRITE_SIZE"][k] / max(cnt["WRITE_SIZE"][k], 1) * 1024.0
    per_pass_us = dur[k] / passes
    per_pass_b = (fb + wb) * calls[k] / passes
    tot_us += per_pass_us
    tot_b += per_pass_b
    print("  %-28s %5d launches  mean %9.1f us  per pass %9.1f us  HBM per launch %8.1f MB read %8.1f MB written  (%.2f TB/s)" % (
        k, calls[k], us, per_pass_us, fb / 1e6, wb / 1e6, (fb + wb) / us / 1e6))
alg = M * (8.0 * N + 16)
print("  per pass: kernels %.1f us, measured HBM %.2f GB = %.2f x the algorithmic %.2f GB; %.3f of 8 TB/s on the algorithmic bytes" % (
    tot_us, tot_b / 1e9, tot_b / alg, alg / 1e9, alg / (tot_us * 1e-6) / 8e12))
