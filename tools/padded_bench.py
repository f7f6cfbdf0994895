#!/usr/bin/env python3
"""N < n = 4096 (leading zero pad): the default kernel with the indicator-correlation correction (auto)
against kernel 7 (statistics barrier + per-sample mean removal)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("go-muse_amd")
eng = pkg.get_engine(0)
for N in (2100, 3000, 4000, 4096):
    rows = 400_000
    dg, ref = pkg.DeviceGroup.synthetic(eng, rows, N)
    db = pkg.DeviceBatch(eng, dg, ref)
    out = []
    for variant in (0, 7):
        eng.set_kernel(variant)
        db.score(); eng.synchronize()
        eng.kernel_timing(True)
        for _ in range(5):
            db.score()
        eng.synchronize()
        ms, cnt = eng.kernel_time()
        eng.kernel_timing(False)
        t = ms / cnt * 1e-3
        out.append("%s %.3f ms (%.1f%% of 8 TB/s)" % ("auto" if variant == 0 else "kernel 7", t * 1e3, rows * (8 * N + 16) / t / 8e12 * 100))
    eng.set_kernel(0)
    print("N=%d: %s" % (N, " | ".join(out)), flush=True)
    db.close(); dg.close()
