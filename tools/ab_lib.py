#!/usr/bin/env python3
"""A/B of two builds of libmuse_hip.so on ONE box (boxes differ by +- 3-5 %): tools/ab_lib.py <other.so> [N ...]
times the all-scores pass (automatic kernel selection, 400 000 rows or 4 GB) per length with the in-tree library and with
<other.so> (e.g. the previous commit's build, copied to tools/ablate/ab_prev/), alternating, each in its own process."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import importlib, os, sys
sys.path.insert(0, %(root)r)
pkg = importlib.import_module("go-muse_amd")
lib = %(lib)r
if lib:
    import ctypes
    pkg.build.LIB = lib
    pkg.build.stale = lambda: False
    L = ctypes.CDLL(lib)
    pkg.binding.SIGNATURES = {k: v for k, v in pkg.binding.SIGNATURES.items() if hasattr(L, k)}   # (an older ABI lacks newer entry points)
eng = pkg.get_engine(0)
for N in %(Ns)r:
    rows = max(2048, min(400_000, (1 << 32) // (8 * N)))
    dg, ref = pkg.DeviceGroup.synthetic(eng, rows, N)
    db = pkg.DeviceBatch(eng, dg, ref)
    db.score(); eng.synchronize()
    eng.kernel_timing(True)
    for _ in range(5):
        db.score()
    eng.synchronize()
    ms, cnt = eng.kernel_time()
    eng.kernel_timing(False)
    t = ms / cnt * 1e-3
    print("%%s N=%%5d rows=%%6d %%8.3f ms %%5.1f%%%% of 8 TB/s" %% (%(tag)r, N, rows, t * 1e3, rows * (8 * N + 16) / t / 8e12 * 100), flush=True)
    db.close(); dg.close()
'''


def main():
    other = os.path.abspath(sys.argv[1])
    Ns = [int(a) for a in sys.argv[2:]] or [3000, 4000, 5000, 10000, 20000, 40000]
    for rnd in range(2):
        for tag, lib in (("in-tree", ""), ("other  ", other)):
            subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "lib": lib, "Ns": Ns, "tag": tag}], check=True)


if __name__ == "__main__":
    main()
