#!/bin/bash
# Resident workgroups per CU of the long-series kernel (xcorr_long.hip): rebuilds it with MUSE_LONG_WGS_PER_CU = $1 ... into the
# library and times tools/sizes_bench.py; run on the GPU box (the box's copy of the library is scratch)
set -e
cd "$(dirname "$0")/../.."
python3 -c "import importlib; importlib.import_module('go-muse_amd.build').build()"
OBJ=go-muse_amd/lib/obj
for w in "$@"; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Igo-muse_amd/csrc -DMUSE_LONG_WGS_PER_CU=$w -c go-muse_amd/csrc/xcorr_long.hip -o $OBJ/xcorr_long.hip.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJ/*.o -o go-muse_amd/lib/libmuse_hip.so
    echo "== MUSE_LONG_WGS_PER_CU=$w"
    SIZES_AUTO_ONLY=1 python3 tools/sizes_bench.py 4000000000 32768 65536 40000
done
