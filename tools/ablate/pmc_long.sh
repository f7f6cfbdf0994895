#!/bin/bash
# HBM-side traffic (FETCH_SIZE, WRITE_SIZE: separate passes) and L2 hit / miss counts of the long-series kernels: the default
# four-step kernel and variant 14 (xcorr_long_team.hip) on one geometry.  usage: tools/pmc_long.sh <tag> [N] [geoms]
set -o pipefail
TAG=${1:-r04}; N=${2:-65536}; export LONG_TEAM_GEOMS=${3:-4x8}
OUT=gpurun_out/pmc_long_$TAG
mkdir -p $OUT
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
    name=$(echo $pass | tr ' ' '_')
    rocprofv3 --pmc $pass --output-format csv -d $OUT/$name -- python3 tools/long_team_bench.py 4294967296 $N > $OUT/$name.log 2>&1 || { tail -5 $OUT/$name.log; exit 1; }
done
python3 - "$OUT" <<'PY' | tee $OUT/summary.txt
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "xcorr_fused_long" in k or "xcorr_long_team" in k:
            acc[k.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    print(k)
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    for c, v in sorted(m.items()):
        print("   %-22s mean per launch %.4g  (%d launches)" % (c, v, len(cs[c])))
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        print("   HBM-side bytes per launch = (2 FETCH_SIZE + WRITE_SIZE) KB = %.4g" % ((2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024))
    if "TCC_HIT_sum" in m:
        print("   L2 hit rate %.3f" % (m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])))
PY
