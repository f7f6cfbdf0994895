#!/bin/bash
# Resident workgroups of the four-step kernel (n >= 32768): rebuilds xcorr_stockham.hip with MUSE_4STEP_WGS_PER_8CU = $1 ...
# into the library and times tools/sizes_bench.py; run on the GPU box (the box's copy of the library is scratch)
set -e
cd "$(dirname "$0")/../.."
python3 -c "import importlib; importlib.import_module('go-muse_amd.build').build(force=True)"
OBJ=go-muse_amd/lib/obj
for w in "$@"; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Igo-muse_amd/csrc -DMUSE_4STEP_WGS_PER_8CU=$w -c go-muse_amd/csrc/xcorr_stockham.hip -o $OBJ/xcorr_stockham.hip.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJ/*.o -o go-muse_amd/lib/libmuse_hip.so
    echo "== MUSE_4STEP_WGS_PER_8CU=$w"
    SIZES_AUTO_ONLY=1 python3 tools/sizes_bench.py 4000000000 32768 65536 40000
done
