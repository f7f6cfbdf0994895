#!/bin/bash
# A/B builds of xcorr_r16_fold.hip's scheduling options: tools/ablate/fold_opt.sh "<MUSE_FOLD_OPT values>" [extra -D flags]
set -e
cd "$(dirname "$0")/../.."
for o in $1; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMUSE_FOLD_OPT=$o $2 -Iinclude -Igo-muse_amd/csrc tools/ablate/fold_phases.hip -o tools/ablate/fold_ablate_opt$o$3 2>&1 | grep -v "warning\|^$" || true
done
