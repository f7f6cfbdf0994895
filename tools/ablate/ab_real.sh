#!/bin/bash
# A/B of compile-time switches of xcorr_real.hip on ONE box: tools/ablate/ab_real.sh "<flags A>" "<flags B>" ... ; recompiles that one
# source per variant, relinks the library, checks parity (tools/real_debug.py) and times test hook 14 (tools/sizes_bench.py)
set -e
cd "$(dirname "$0")/../.."
LIB=go-muse_amd/lib
for round in 1 2; do
  for flags in "$@"; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Igo-muse_amd/csrc $flags -c go-muse_amd/csrc/xcorr_real.hip -o $LIB/obj/xcorr_real.hip.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $LIB/obj/*.o -o $LIB/libmuse_hip.so
    echo "== round $round flags '$flags'"
    [ $round = 1 ] && python3 tools/real_debug.py ${AB_CHECK:-32768 20000 24577} | grep "variant 14"
    SIZES_AUTO_ONLY=1 SIZES_VARIANT=14 python3 tools/sizes_bench.py 4294967296 ${AB_SIZES:-32768 20000 24001}
  done
done
