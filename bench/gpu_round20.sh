#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH=$PWD
for nb in 1024 512 768; do
PDT_WGRAD_BLOCKS=$nb timeout -k 10 300 python -u bench/resnet50_roofline.py --batch 256 > gpurun_out/roofline_nb$nb.md 2>&1
echo "nb=$nb: $(grep 'conv_wgrad     measured' gpurun_out/roofline_nb$nb.md)"
PDT_WGRAD_BLOCKS=$nb timeout -k 10 300 python -u bench.py --steps 10 --warmup 4 > gpurun_out/bench_nb$nb.log 2>&1
echo "b1024 nb=$nb: $(tail -1 gpurun_out/bench_nb$nb.log | cut -c1-160)"
done
exit 0
