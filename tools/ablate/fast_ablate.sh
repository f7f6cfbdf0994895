#!/bin/bash
# Ceilings of the default n = 4096 kernel (xcorr_r16_fast.hip): the stamped harness built with parts removed.
#   MUSE_ABLATE=1  no LDS transposes, no workgroup barriers   (arithmetic + global reads)
#   MUSE_ABLATE=2  rows re-read from L2 (no HBM traffic)      (arithmetic + LDS + barriers)
#   MUSE_ABLATE=3  both: arithmetic + L2-hot loads            (the VALU floor at the clock the chip holds)
set -e
cd "$(dirname "$0")/../.."
for a in 0 1 2 3; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMUSE_ABLATE=$a -Iinclude -Igo-muse_amd/csrc tools/ablate/fast_phases.hip -o tools/ablate/fast_ablate_$a
done
