#!/bin/bash
# Per-kernel time and measured HBM traffic of the long-series pass (xcorr_huge.hip) at given lengths: rocprofv3 --kernel-trace
# --stats, then FETCH_SIZE and WRITE_SIZE in passes of their own (gpurun refuses counters combined with tracing), over
# tools/huge_bench.py; 2 x FETCH_SIZE + WRITE_SIZE as MI355X_MICROARCH.md prescribes for gfx950.
#   usage (GPU box, repo root): tools/profile_huge.sh <out dir> <group GB> N [N ...]
set -o pipefail
OUT=$1; shift
GB=$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
for N in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$N -- python3 tools/huge_bench.py $GB $N > $OUT/trace_$N.log 2>&1 || { tail -5 $OUT/trace_$N.log; exit 1; }
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$N -- python3 tools/huge_bench.py $GB $N > $OUT/fetch_$N.log 2>&1 || { tail -5 $OUT/fetch_$N.log; exit 1; }
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write_$N -- python3 tools/huge_bench.py $GB $N > $OUT/write_$N.log 2>&1 || { tail -5 $OUT/write_$N.log; exit 1; }
  python3 tools/profile_huge_summary.py $OUT $N $GB
done
