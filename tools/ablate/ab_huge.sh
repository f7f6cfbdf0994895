#!/bin/bash
# Where sweep 1 of the long-series pass goes: a diagnostic build (-DMUSE_HUGE_ABL) of xcorr_huge.hip / capi_huge.hip whose sweep 1
# leaves parts out at run time (MUSE_HUGE_ABL bits: 1 = no W_n twiddles, 2 = contiguous stores, 4 = contiguous row reads, 8 = no column DFT at all (load -> store), 16 = no butterflies, 32 = no W_R1 factors), timed per
# kernel with rocprofv3 --kernel-trace.  Results are WRONG under any bit; the product build has none of this.
#   usage (GPU box, repo root): tools/ablate/ab_huge.sh <out dir> N
set -o pipefail
OUT=$1; N=$2
mkdir -p $OUT
export TMPDIR=/tmp
cp go-muse_amd/lib/libmuse_hip.so $OUT/libmuse_hip.product.so
python3 go-muse_amd/build.py --force -DMUSE_HUGE_ABL $ABL_FLAGS > $OUT/build.log 2>&1 || { tail -5 $OUT/build.log; exit 1; }
for bits in ${ABL_BITS:-0 1 2 4 6 7}; do
  MUSE_HUGE_ABL=$bits rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/abl_$bits -- python3 tools/huge_bench.py 2 $N > $OUT/abl_$bits.log 2>&1 || { tail -5 $OUT/abl_$bits.log; }
  echo "MUSE_HUGE_ABL=$bits: $(grep -h huge_sweep1 $OUT/abl_$bits/*/*kernel_stats.csv | head -1 | cut -d, -f1-5)"
done
cp $OUT/libmuse_hip.product.so go-muse_amd/lib/libmuse_hip.so
