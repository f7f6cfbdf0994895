#!/bin/bash
# Batch size of the batched long-series path (xcorr_long_batched.hip): rebuilds muse_capi.hip with MUSE_LONG_BATCH_MB = $1 ... and
# times test-hook kernel 14 with tools/sizes_bench.py; run on the GPU box (its library copy is scratch)
set -e
cd "$(dirname "$0")/../.."
python3 -c "import importlib; importlib.import_module('go-muse_amd.build').build()"
OBJ=go-muse_amd/lib/obj
for w in "$@"; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Igo-muse_amd/csrc -DMUSE_LONG_BATCH_MB=$w -c go-muse_amd/csrc/muse_capi.hip -o $OBJ/muse_capi.hip.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJ/*.o -o go-muse_amd/lib/libmuse_hip.so
    echo "== MUSE_LONG_BATCH_MB=$w"
    SIZES_AUTO_ONLY=1 SIZES_VARIANT=14 python3 tools/sizes_bench.py 16000000000 32768 65536
done
