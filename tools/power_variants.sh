#!/bin/bash
# Clock / package power while each n = 4096 kernel variant loops (tools/kbench.py), sampled with rocm-smi.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for v in 10 7 8 5 2; do
    timeout -k 10 200 python tools/kbench.py 1000000 ${ROUNDS:-1500} $v > gpurun_out/pv_$v.log 2>&1 &
    pid=$!
    sleep ${WARM:-14}
    s=""
    for i in 1 2 3; do
        c=$(rocm-smi --showclocks 2>/dev/null | grep sclk | sed -E 's/.*\(([0-9]+)Mhz\).*/\1/')
        w=$(rocm-smi --showpower 2>/dev/null | grep -i "Power (W)" | sed -E 's/.*: *([0-9.]+).*/\1/')
        s="$s ${c}MHz/${w}W"
        sleep 1
    done
    wait $pid
    echo "variant $v:$s  |  $(tail -1 gpurun_out/pv_$v.log)"
done
