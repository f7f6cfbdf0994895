#!/bin/bash
# rocprofv3 per-kernel averages for the config-5 lengths (tools/sizes_bench.py, automatic kernel selection only) plus the
# same script's own event timing; run on the GPU box from the repo root.  usage: tools/profile_sizes.sh [tag]
set -o pipefail
TAG=${1:-r02}
OUT=gpurun_out/prof_sizes_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export SIZES_AUTO_ONLY=1
LENGTHS="480 512 1000 1024 1500 2048 4096 5000 8192 10000 16384 20000 32768 40000 65536"
python3 tools/sizes_bench.py 8000000000 $LENGTHS > $OUT/events.txt 2>&1 || { tail -5 $OUT/events.txt; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/sizes_bench.py 8000000000 $LENGTHS > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
{
    echo "# tools/profile_sizes.sh $TAG: config-5 lengths, float64, automatic kernel selection, ~8 GB groups (rows capped at 400 000)"
    echo "# (a) the script's own HIP-event timing of the all-scores pass: fraction of the 8 TB/s roofline on 8 N + 16 bytes per series"
    cat $OUT/events.txt
    echo
    echo "# (b) rocprofv3 --kernel-trace --stats of the same command: per-kernel calls / average ns (one warm-up + three timed passes per length)"
    python3 - "$OUT" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "xcorr_fused" in r["Name"]:
            print("%-90s calls=%-4s avg_ns=%-12s total_ns=%s" % (r["Name"][:90], r["Calls"], r["AverageNs"], r["TotalDurationNs"]))
PY
} > $OUT/summary.txt
cat $OUT/summary.txt
