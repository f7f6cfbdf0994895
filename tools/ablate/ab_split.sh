# A/B on one box: n = 32768 with the 16384-point transforms as 16 x 1024 (test hook 15 = automatic selection since) against the three-transpose form (hook 14)
set -e
for v in 14 15 14 15; do SIZES_VARIANT=$v SIZES_AUTO_ONLY=1 timeout -k 10 200 python tools/sizes_bench.py 4294967296 20000 24001 32768; done
