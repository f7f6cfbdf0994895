#!/bin/bash
# Where a row's time goes in xcorr_fused_real32k (test hook 14) and its 16 x 1024 form (hook 15): builds with parts of the iteration
# left out (MUSE_REAL_ABL bits: 1 row requests, 2 mirror stage, 4 second transform, 8 first transform, 16 statistics reduction), one box.
# The results of such a build are wrong; only its time is read.  The last build is the full kernel again.
set -e
cd "$(dirname "$0")/../.."
LIB=go-muse_amd/lib
for abl in ${AB_LIST:-0 1 2 4 8 16 14 31 0}; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Igo-muse_amd/csrc -DMUSE_REAL_ABL=$abl -c go-muse_amd/csrc/xcorr_real.hip -o $LIB/obj/xcorr_real.hip.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $LIB/obj/*.o -o $LIB/libmuse_hip.so
    for v in 14 15; do
        echo "== ABL $abl hook $v"
        SIZES_AUTO_ONLY=1 SIZES_VARIANT=$v timeout -k 10 120 python3 tools/sizes_bench.py 4294967296 32768
    done
done
