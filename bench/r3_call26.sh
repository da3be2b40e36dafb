#!/bin/bash
# ViT-B/16: where does the per-image time go at B = 2560 against B = 1024
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out/r3 && export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
for b in 1024 2560; do
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r3/vit_prof_b$b -o vit -- python3 $R/bench.py --model vit_b_16 --optimizer adamw --batch_size $b --steps 4 --warmup 2 > $R/gpurun_out/r3/vit_prof_b$b.log 2>&1 || exit 1
grep '^{' $R/gpurun_out/r3/vit_prof_b$b.log | cut -c1-170
done
