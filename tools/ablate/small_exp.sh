#!/bin/bash
# Table-latency ablations of the half-round kernel: library builds with MUSE_SMALL_EXP = $1 ... (results are wrong, times are what
# is left), timed with tools/sizes_bench.py on ONE box; the last build restores the library
set -e
cd "$(dirname "$0")/../.."
for w in "$@"; do
    python3 go-muse_amd/build.py -DMUSE_SMALL_EXP=$w > /dev/null 2>&1
    echo "== MUSE_SMALL_EXP=$w"
    SIZES_AUTO_ONLY=1 python3 tools/sizes_bench.py 8000000000 ${SIZES_LIST:-512 1024 2048 8192}
done
python3 go-muse_amd/build.py > /dev/null 2>&1
