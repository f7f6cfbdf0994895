#!/bin/bash
# rocprofv3 kernel statistics of bench.py including the many-references object (run on the GPU box from the repo root).
set -o pipefail
OUT=gpurun_out/prof_${1:-many}
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --many-refs ${2:-8} > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -16 "$f" | cut -c1-200 > $OUT/kernel_stats_head.csv && cat $OUT/kernel_stats_head.csv
