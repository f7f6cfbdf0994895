#!/bin/bash
# Extra SQ counter passes for one fused-kernel variant (diagnostic): usage tools/pmc_probe.sh <tag> <variant>
set -o pipefail
TAG=${1:-probe}; VAR=${2:-10}
OUT=/root/repo/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /root/repo
export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_IFETCH SQ_IFETCH_LEVEL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL --output-format csv -d $OUT/pmc_sq1 -- python3 tools/kbench.py 1000000 2 $VAR > $OUT/p1.log 2>&1 || { tail -5 $OUT/p1.log; exit 1; }
rocprofv3 --pmc SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU --output-format csv -d $OUT/pmc_sq2 -- python3 tools/kbench.py 1000000 2 $VAR > $OUT/p2.log 2>&1 || { tail -5 $OUT/p2.log; exit 1; }
python3 tools/profile_summary.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
