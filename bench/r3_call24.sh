#!/bin/bash
# ViT-B/16 at larger per-GPU batches + kernel stats
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out/r3 && export PYTHONPATH=$R
for b in 110 220 332 512 1024; do
  echo "== vit B=$b"
  timeout -k 10 300 python -u bench.py --model vit_b_16 --optimizer adamw --batch_size $b --steps 15 --warmup 4 2>/dev/null | cut -c1-180 || exit 1
done
echo "== vit default batch"
timeout -k 10 400 python -u bench.py --model vit_b_16 --optimizer adamw --steps 10 --warmup 3 2>/dev/null > gpurun_out/r3/vit_default.json || exit 1
cut -c1-180 gpurun_out/r3/vit_default.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r3/vit_prof -o vit512 -- python3 $R/bench.py --model vit_b_16 --optimizer adamw --batch_size 512 --steps 6 --warmup 2 > /dev/null 2>&1 || exit 1
ls $R/gpurun_out/r3/vit_prof | head
