#!/usr/bin/env python3
"""Shader clock held under the two-sided n = 4096 kernel against the one-sided (headline) kernel, same process, same box
(is the two-sided kernel's 64 % vector-busy figure real, or an artefact of a lower clock?).
The probe wave of muse_test_clock_probe_* samples the clock in 1 ms windows beside a burst of each; the median over the
windows past the first 40 ms of the burst (the ramp) is printed next to the kernel's HIP-event time.
usage: clock_trace_two_sided.py [pairs] [bursts]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("go-muse_amd")
P = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
bursts = int(sys.argv[2]) if len(sys.argv) > 2 else 3
N = 4096
eng = pkg.get_engine(0)
gx, ref = pkg.DeviceGroup.synthetic(eng, P, N, seed=0x78636F72)
gy, _ = pkg.DeviceGroup.synthetic(eng, P, N, seed=0x6D757365)
db = pkg.DeviceBatch(eng, gx, ref)


def burst(name, call, calls):
    for _ in range(4):
        call()
    eng.synchronize()
    eng.kernel_time()
    t_est = 0.0
    t0 = time.perf_counter()
    call()
    eng.synchronize()
    t_est = (time.perf_counter() - t0) * 1e3
    total = 50 + calls * t_est * 1.1 + 50
    eng.clock_probe_start(1.0, total)
    time.sleep(0.03)
    eng.kernel_timing(True)
    t0 = time.perf_counter()
    for _ in range(calls):
        call()
    eng.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    eng.kernel_timing(False)
    ms, cnt = eng.kernel_time()
    eng.clock_probe_stop()
    mhz = np.asarray(eng.clock_probe_read())
    lo, hi = 30 + 60, int(30 + min(dt, ms) * 0.85)   # windows inside the burst of kernels (ms = their summed HIP-event time), past its first 60 ms
    inside = mhz[lo:hi] if hi > lo + 5 else mhz
    print("%-34s %3d calls in %7.1f ms, kernel avg %7.3f ms; clock inside the burst: median %5.0f MHz (p10 %5.0f, p90 %5.0f, %d windows)" % (
        name, calls, dt, ms / max(cnt, 1), np.median(inside), np.percentile(inside, 10), np.percentile(inside, 90), len(inside)), flush=True)
    return np.median(inside), ms / max(cnt, 1)


# (one muse_xcorr_groups call costs the host tens of milliseconds around its kernel -- output buffers, copies back, the NaN rescue
# pass -- so calls in a loop leave the GPU idle most of the time and the clock at its 2.4 GHz boost: the measurement hook makes ONE
# call launch the kernel `REPEAT` times back to back instead)
REPEAT = 150
for b in range(bursts):
    c1, k1 = burst("one-sided xcorr_fused_n4096_fold", lambda: db.score(), 120)
    eng.xcorr_repeat(REPEAT)
    c2, k2 = burst("two-sided normalize=1 (x%d per call)" % REPEAT, lambda: pkg.xcorr_groups(gx, gy, N, True), 1)
    c3, k3 = burst("two-sided normalize=0 (x%d per call)" % REPEAT, lambda: pkg.xcorr_groups(gx, gy, N, False), 1)
    eng.xcorr_repeat(1)
    print("  burst %d: clock ratio two-sided / one-sided = %.3f (normalized), %.3f (raw); two-sided kernel at that clock: %.3f ms = %.1f %% of 8 TB/s on 16 N + 16 B per pair" % (
        b, c2 / c1, c3 / c1, k2, P * (16.0 * N + 16) / (k2 * 1e-3) / 8e12 * 100), flush=True)
