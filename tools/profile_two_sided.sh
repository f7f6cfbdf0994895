#!/bin/bash
# HBM traffic of the batched two-sided xCorr kernels per FFT length: rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in runs of their own over
# tools/two_sided_bench.py (run on the GPU box from the repo root); bytes per launch = FETCH_SIZE x 2 + WRITE_SIZE in KB x 1024 as
# MI355X_MICROARCH.md prescribes for gfx950 (tools/profile_summary.py applies the same correction).  usage: tools/profile_two_sided.sh [tag] [N ...]
set -o pipefail
TAG=${1:-r05}; shift
LENGTHS=${@:-4096 8192 16384 32768 65536}
OUT=gpurun_out/prof_two_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $OUT/$c -- python3 tools/two_sided_bench.py 200000 $LENGTHS > $OUT/$c.log 2>&1 || { tail -5 $OUT/$c.log; exit 1; }
done
python3 - "$OUT" $LENGTHS <<'PY' | tee $OUT/summary.txt
import csv, glob, sys, collections
out = sys.argv[1]
acc = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(out + "/" + c + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "two_sided" in r["Kernel_Name"] and r["Counter_Name"] == c:
                k = r["Kernel_Name"].split("(")[0]
                a = acc.setdefault(k, collections.defaultdict(list))
                a[c].append(float(r["Counter_Value"]))
log = open(out + "/FETCH_SIZE.log").read().splitlines()
print("# tools/profile_two_sided.sh: HBM bytes per launch of the two-sided kernels (2 x FETCH_SIZE + WRITE_SIZE, KB -> bytes), means over each kernel's launches")
for k, a in sorted(acc.items()):
    f = sum(a["FETCH_SIZE"]) / max(len(a["FETCH_SIZE"]), 1)
    w = sum(a["WRITE_SIZE"]) / max(len(a["WRITE_SIZE"]), 1)
    print("%-60s launches %3d  bytes per launch %.4e" % (k[:60], len(a["FETCH_SIZE"]), (2 * f + w) * 1024))
print("# the tool's own lines (pairs per launch and the algorithmic 16 N + 16 bytes per pair are in them):")
for l in log:
    if l.startswith("N="):
        print(l)
PY
