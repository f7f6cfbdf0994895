#!/usr/bin/env python3
"""Debug aid: screened Run vs fp64 Run on the synthetic rows; prints the screening pass's statistics."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("go-muse_amd")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
eng = pkg.get_engine(0)
dg, ref = pkg.DeviceGroup.synthetic(eng, M, 4096, seed=99)
db = pkg.DeviceBatch(eng, dg, ref)
lag, mv = db.scores()
args = (None, 0, 15, 20, 0.0, 0, True)
eng.set_screening(False)
exp = db.run(*args)
eng.set_screening(True)
got = db.run(*args)
print("exp", exp[0][:8], exp[1][:8], exp[2][:8])
print("got", got[0][:8], got[1][:8], got[2][:8])
print("exact of got rows", mv[got[0][:8]], lag[got[0][:8]])
est, flags, E = db.screen_estimates(15)
ref_ = (flags >> 31) & 1
print("E", E, "refined rows", int(ref_.sum()), "of", M)
for bit, name in ((1, "IN"), (2, "OUT"), (4, "POS"), (8, "NEG"), (16, "REFINE"), (32, "NAN")):
    print(name, int(((flags & bit) != 0).sum()))
chk = ref_ == 0
err = np.abs(np.abs(est[chk]) - np.abs(mv[chk]))
print("max err / E", err.max() / E, "argmax row", np.nonzero(chk)[0][err.argmax()])
inside = np.abs(lag) <= 15
print("rows inside", int(inside.sum()), "flag IN only", int((((flags & 3) == 1)).sum()), "IN|OUT", int(((flags & 3) == 3).sum()))
top = np.argsort(-np.abs(np.where(inside, mv, 0)))[:25]
print("top exact:", top[:10], mv[top[:10]], lag[top[:10]], flags[top[:10]] & 63, est[top[:10]])
