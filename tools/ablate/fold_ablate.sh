#!/bin/bash
# Ceilings of the default n = 4096 kernel (xcorr_r16_fold.hip): the harness built with parts removed.
#   MUSE_FOLD_EXP bit 0: pass-3 factors from LDS instead of L2;  bit 1: spectrum factors from LDS;  bit 3: no result write-out
#   MUSE_ABLATE  bit 1: rows re-read from L2 (no HBM traffic)
set -e
cd "$(dirname "$0")/../.."
for cfg in "0 0" "1 0" "2 0" "3 0" "8 0" "0 2" "3 2" "11 2"; do
    set -- $cfg
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMUSE_FOLD_EXP=$1 -DMUSE_ABLATE=$2 -Iinclude -Igo-muse_amd/csrc tools/ablate/fold_phases.hip -o tools/ablate/fold_ablate_$1_$2
done
