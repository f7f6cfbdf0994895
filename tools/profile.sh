#!/bin/bash
# Collects the rocprofv3 evidence for bench.py's roofline line (run on the GPU box):
#   1) --kernel-trace --stats  -> per-kernel average duration
#   2) --pmc passes (separately): FETCH_SIZE ; WRITE_SIZE ; SQ counters
# Summaries are written under gpurun_out/prof_<tag>/ ; copy the ones to keep into profiles/.
set -o pipefail
TAG=${1:-r02}
STEPS=${2:-5}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /root/repo
export TMPDIR=/tmp
ARGS="bench.py --steps $STEPS --warmup 1 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1 || { tail -5 $OUT/pmc_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.log 2>&1 || { tail -5 $OUT/pmc_write.log; exit 1; }
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq1 -- python3 $ARGS > $OUT/pmc_sq1.log 2>&1 || { tail -5 $OUT/pmc_sq1.log; exit 1; }
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d $OUT/pmc_sq2 -- python3 $ARGS > $OUT/pmc_sq2.log 2>&1 || { tail -5 $OUT/pmc_sq2.log; exit 1; }
python3 tools/profile_summary.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
