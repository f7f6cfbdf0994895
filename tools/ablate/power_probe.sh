#!/bin/bash
# Clock and package power while the default n = 4096 kernel (xcorr_r16_fold.hip harness) and its ablation builds loop
# (needs rocm-smi on the GPU box; build the binaries first: tools/ablate/fold_ablate.sh).
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for cfg in 0_0 3_0 0_2 11_2; do
    REPS=700 NOSTAMP=1 timeout -k 10 120 ./tools/ablate/fold_ablate_$cfg > gpurun_out/abl_$cfg.log 2>&1 &
    pid=$!
    sleep 5
    s=""
    for i in 1 2 3; do
        c=$(rocm-smi --showclocks 2>/dev/null | grep sclk | sed -E 's/.*\(([0-9]+)Mhz\).*/\1/' | head -1)
        w=$(rocm-smi --showpower 2>/dev/null | grep -i "Power (W)" | sed -E 's/.*: *([0-9.]+).*/\1/' | head -1)
        s="$s ${c}MHz/${w}W"
        sleep 1
    done
    wait $pid
    echo "MUSE_FOLD_EXP_MUSE_ABLATE=$cfg (sampled 5-8 s in):$s"
    grep -E "^FOLD" gpurun_out/abl_$cfg.log | tr '\n' ';'
    echo
done
