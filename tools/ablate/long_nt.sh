#!/bin/bash
# Cache policy of the long-series kernel's scratch slice: rebuilds xcorr_long.hip with MUSE_LONG_NT = $1 ... (bit 0 stores, bit 1 loads
# non-temporal) into the library and times tools/sizes_bench.py; run on the GPU box (its library copy is scratch)
set -e
cd "$(dirname "$0")/../.."
python3 -c "import importlib; importlib.import_module('go-muse_amd.build').build()"
OBJ=go-muse_amd/lib/obj
for w in "$@"; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Igo-muse_amd/csrc -DMUSE_LONG_NT=$w -c go-muse_amd/csrc/xcorr_long.hip -o $OBJ/xcorr_long.hip.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJ/*.o -o go-muse_amd/lib/libmuse_hip.so
    echo "== MUSE_LONG_NT=$w"
    SIZES_AUTO_ONLY=1 python3 tools/sizes_bench.py 16000000000 32768 65536
done
