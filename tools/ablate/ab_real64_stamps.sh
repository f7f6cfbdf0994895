#!/bin/bash
# Where a series of n = 65536 goes inside xcorr_fused_real64k: a diagnostic build (-DMUSE_REAL64_STAMPS) stamps the
# shader clock at eleven points of the series loop, per wave, summed over the series a workgroup handles; the context dumps the sums
# when it is destroyed (MUSE_STAMPS_OUT).  The stamps cost a sched_barrier each and one s_waitcnt: the un-stamped time is printed next
# to the stamped one.  usage (GPU box, repo root): tools/ablate/ab_real64_stamps.sh <out dir> [N]
set -o pipefail
OUT=$1; N=${2:-65536}
mkdir -p $OUT
cp go-muse_amd/lib/libmuse_hip.so $OUT/libmuse_hip.product.so
echo "== product build" > $OUT/stamps.txt
SIZES_AUTO_ONLY=1 python3 tools/sizes_bench.py 4294967296 $N >> $OUT/stamps.txt 2>&1
python3 go-muse_amd/build.py --force -DMUSE_REAL64_STAMPS > $OUT/build.log 2>&1 || { tail -5 $OUT/build.log; exit 1; }
echo "== stamped build" >> $OUT/stamps.txt
MUSE_STAMPS_OUT=$OUT/stamps.raw SIZES_AUTO_ONLY=1 python3 tools/sizes_bench.py 4294967296 $N >> $OUT/stamps.txt 2>&1
cp $OUT/libmuse_hip.product.so go-muse_amd/lib/libmuse_hip.so
python3 - $OUT/stamps.raw >> $OUT/stamps.txt <<'PY'
import sys
import numpy as np
rows = np.loadtxt(sys.argv[1], dtype=np.float64)
acc = rows[:, 2:13]
live = acc.sum(axis=1) > 0
acc = acc[live]
names = ["rows requested + consumed, V parked", "statistics reduced (2 barriers)", "transform A1", "mirror stage A", "transform A2",
         "ce parked, V read back (waited)", "transform B1", "mirror stage B", "transform B2", "ce read back + combine + lane maximum",
         "workgroup maximum, result, closing barrier"]
tot = acc.sum(axis=1)
print("waves with stamps: %d; cycles per wave (sum over its series): mean %.3e" % (len(acc), tot.mean()))
share = acc / tot[:, None]
for i, nm in enumerate(names):
    print("  %5.1f %%  (min %4.1f, max %4.1f over the waves)  %s" % (100 * share[:, i].mean(), 100 * share[:, i].min(), 100 * share[:, i].max(), nm))
mem = share[:, [0, 5, 9]].sum(axis=1).mean()
print("  memory-facing phases (0, 5, 9): %.1f %%; transforms + mirror stages: %.1f %%; reductions / barriers: %.1f %%" % (
    100 * mem, 100 * share[:, [2, 3, 4, 6, 7, 8]].sum(axis=1).mean(), 100 * share[:, [1, 10]].sum(axis=1).mean()))
PY
cat $OUT/stamps.txt
