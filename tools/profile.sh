#!/bin/bash
# Collects the rocprofv3 evidence behind EVERY object of bench.py's line (run on the GPU box, from the repo root):
#   1) --kernel-trace --stats  -> per-kernel average duration
#   2) --pmc passes, each in a run of its own (gpurun refuses counters combined with tracing): FETCH_SIZE ; WRITE_SIZE ; SQ counters
# over the default bench workload WITH its extra objects (float32-storage group, many references, two-sided xCorr, config-5
# lengths), then tools/profile_summary.py condenses them into summary.txt + counters.json (per kernel instantiation: HBM bytes
# per launch = 2 * FETCH_SIZE + WRITE_SIZE as MI355X_MICROARCH.md prescribes for gfx950, vector / LDS instruction and cycle counts).
# Copy gpurun_out/prof_<tag>/counters.json to profiles/<tag>_counters.json: bench.py attaches `traffic` and `co_bounds` from the
# newest such file and refuses it when the kernel sources have changed since (csrc_sha).
#   usage: tools/profile.sh <tag> [steps] ; PROFILE_COMMIT=<git rev> is recorded in counters.json
set -o pipefail
TAG=${1:-r03}
STEPS=${2:-4}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
ARGS="bench.py --steps $STEPS --warmup 1 --no-cpu-baseline --skip-extra in_process_shards --skip-extra reference_bench_shapes"
run() { # name, rocprofv3 options...
    local name=$1; shift
    echo "== $name: rocprofv3 $*" >&2
    rocprofv3 "$@" --output-format csv -d $OUT/$name -- python3 $ARGS > $OUT/$name.log 2>&1 || { tail -5 $OUT/$name.log; exit 1; }
}
ARGS_PMC=$ARGS
ARGS="bench.py --steps 16 --warmup 3 --no-cpu-baseline --skip-extra in_process_shards --skip-extra reference_bench_shapes"   # (the trace pass: enough launches for its
run trace --kernel-trace --stats                                                       #  average to sit behind the 30 ms clock ramp)
ARGS=$ARGS_PMC
run pmc_fetch --pmc FETCH_SIZE
run pmc_write --pmc WRITE_SIZE
run pmc_sq1 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run pmc_sq2 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_SALU
python3 tools/profile_summary.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
