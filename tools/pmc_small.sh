#!/bin/bash
# SQ counters of the per-length kernels (tools/sizes_bench.py for the given lengths); run on the GPU box from the repo root
set -o pipefail
OUT=gpurun_out/pmc_small
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq1 -- python3 tools/sizes_bench.py 4294967296 "$@" > $OUT/sq1.log 2>&1 || { tail -5 $OUT/sq1.log; exit 1; }
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d $OUT/sq2 -- python3 tools/sizes_bench.py 4294967296 "$@" > $OUT/sq2.log 2>&1 || { tail -5 $OUT/sq2.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
for sub in ("sq1", "sq2"):
    for f in glob.glob("gpurun_out/pmc_small/%s/**/*counter_collection.csv" % sub, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            if "xcorr_fused" not in k:
                continue
            print(k[:70])
            for c, v in sorted(cs.items()):
                print("   %-26s n=%d mean=%.5g" % (c, len(v), sum(v) / len(v)))
PY
