#!/bin/bash
# Ablations of the team kernel for long series: rebuilds xcorr_long_team.hip with MUSE_TEAM_EXP = $1 ... (bit 0: slice traffic without
# sc1, bit 1: team barriers do not wait -- results are wrong) and times test-hook kernel 14; run on the GPU box
set -e
cd "$(dirname "$0")/../.."
python3 -c "import importlib; importlib.import_module('go-muse_amd.build').build()"
OBJ=go-muse_amd/lib/obj
for w in "$@"; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Igo-muse_amd/csrc -DMUSE_TEAM_EXP=$w -c go-muse_amd/csrc/xcorr_long_team.hip -o $OBJ/xcorr_long_team.hip.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJ/*.o -o go-muse_amd/lib/libmuse_hip.so
    echo "== MUSE_TEAM_EXP=$w"
    SIZES_AUTO_ONLY=1 SIZES_VARIANT=14 timeout -k 10 120 python3 tools/sizes_bench.py 16000000000 32768 65536
done
