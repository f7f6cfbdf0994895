#!/usr/bin/env python3
"""50 Runs over 100 label groups and 50 Runs without a label map over 10 000 x 4096, for a kernel trace:
rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 tools/probe/trace10k.py (from the repo root)"""
import importlib, os, sys, numpy as np
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("go-muse_amd")
eng = pkg.get_engine(0)
dg, ref = pkg.DeviceGroup.synthetic(eng, 10000, 4096)
db = pkg.DeviceBatch(eng, dg, ref)
gid = (np.arange(10000) % 100).astype(np.int32)
for _ in range(50):
    db.run(gid, 100)
for _ in range(50):
    db.run(None, 0)
