#!/bin/bash
# Clock and package power while each ablation build of the default kernel runs (needs rocm-smi on the GPU box).
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for a in 0 2 3 1; do
    LOOPS=700 timeout -k 10 120 ./tools/ablate/fast_ablate_$a > gpurun_out/abl_$a.log 2>&1 &
    pid=$!
    sleep 6
    s=""
    for i in 1 2 3; do
        c=$(rocm-smi --showclocks 2>/dev/null | grep sclk | sed -E 's/.*\(([0-9]+)Mhz\).*/\1/')
        w=$(rocm-smi --showpower 2>/dev/null | grep -i "Power (W)" | sed -E 's/.*: *([0-9.]+).*/\1/')
        s="$s ${c}MHz/${w}W"
        sleep 1
    done
    wait $pid
    echo "MUSE_ABLATE=$a (first configuration of the harness, 3 waves/SIMD, sampled 6-9 s in):$s"
    grep -E "^FAST" gpurun_out/abl_$a.log | tr '\n' ';'
    echo
done
