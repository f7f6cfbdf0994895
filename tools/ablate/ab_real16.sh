# A/B on one box: n = 16384 as one real series per 512-thread workgroup (xcorr_real.hip) against the pair-packed kernel (test hook 12)
set -e
for v in 0 12 0 12; do SIZES_VARIANT=$v SIZES_AUTO_ONLY=1 timeout -k 10 200 python tools/sizes_bench.py 4294967296 10000 12000 16384; done
for v in 0 12 0 12; do MUSE_TEST_KERNEL=$v timeout -k 10 200 python tools/two_sided_bench.py 200000 16384; done
