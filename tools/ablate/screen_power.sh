#!/bin/bash
# Clock / package power while the screening pass loops (tools/ablate/screen_only, LOOPS launches per configuration).
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
LOOPS=${LOOPS:-3000} timeout -k 10 200 tools/ablate/screen_only 1000000 2 > gpurun_out/screen_power.log 2>&1 < /dev/null &
pid=$!
sleep ${WARM:-12}
for i in 1 2 3 4 5 6; do
    c=$(rocm-smi --showclocks 2>/dev/null | grep sclk | sed -E 's/.*\(([0-9]+)Mhz\).*/\1/')
    w=$(rocm-smi --showpower 2>/dev/null | grep -i "Power (W)" | sed -E 's/.*: *([0-9.]+).*/\1/')
    echo "t+$((12 + 2 * i))s: ${c}MHz ${w}W"
    sleep 2
done
wait $pid
grep -v "cycles/pair" gpurun_out/screen_power.log
