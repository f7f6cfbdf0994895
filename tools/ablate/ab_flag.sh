#!/bin/bash
# A/B of one compile-time flag over the whole library on ONE box: tools/ablate/ab_flag.sh "<-D flag>" <sizes_bench args...>
# builds the library without and with the flag (twice each, alternating) and times tools/sizes_bench.py
set -e
cd "$(dirname "$0")/../.."
FLAG=$1; shift
for round in 1 2; do
  for f in "" "$FLAG"; do
    python3 go-muse_amd/build.py $f > /dev/null 2>&1
    echo "== round $round flag '$f'"
    SIZES_AUTO_ONLY=1 python3 tools/sizes_bench.py "$@"
  done
done
python3 go-muse_amd/build.py > /dev/null 2>&1
