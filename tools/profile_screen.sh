#!/bin/bash
# rocprofv3 kernel statistics of bench.py with the filter-and-refine Run enabled (run on the GPU box from the repo root).
set -o pipefail
OUT=gpurun_out/prof_${1:-screen}
mkdir -p $OUT
export TMPDIR=/tmp
export MUSE_HIP_SCREEN_RUN=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps ${2:-5} --warmup 1 --no-cpu-baseline --many-refs 0 > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
tail -1 $OUT/trace.log | cut -c1-400
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -14 "$f" | cut -c1-220 > $OUT/kernel_stats_head.csv && cat $OUT/kernel_stats_head.csv
