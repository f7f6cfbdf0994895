#!/bin/bash
# Quads per workgroup of the long-series kernel: rebuilds xcorr_long.hip with MUSE_LONG_QD = $1 ... into the library, runs the
# parity tests of the long lengths on that build and times tools/sizes_bench.py; run on the GPU box (its library copy is scratch)
set -e
cd "$(dirname "$0")/../.."
python3 -c "import importlib; importlib.import_module('go-muse_amd.build').build()"
OBJ=go-muse_amd/lib/obj
for w in "$@"; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Igo-muse_amd/csrc -DMUSE_LONG_QD=$w -c go-muse_amd/csrc/xcorr_long.hip -o $OBJ/xcorr_long.hip.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJ/*.o -o go-muse_amd/lib/libmuse_hip.so
    echo "== MUSE_LONG_QD=$w"
    timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "stockham and (16384 or 20000 or 32768 or 40000 or 65536)" 2>&1 | tail -2
    SIZES_AUTO_ONLY=1 python3 tools/sizes_bench.py 16000000000 32768 65536 40000
done
