#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH=$PWD
for rep in 1 2; do
for tall in 0 1; do
PDT_IGEMM_TALL=$tall timeout -k 10 300 python -u bench.py --steps 12 --warmup 4 > gpurun_out/bench_tall$tall.log 2>&1
echo "b1024 tall=$tall: $(tail -1 gpurun_out/bench_tall$tall.log | cut -c1-160)"
PDT_IGEMM_TALL=$tall timeout -k 10 300 python -u bench.py --steps 20 --warmup 5 --batch_size 256 > gpurun_out/bench256_tall$tall.log 2>&1
echo "b256 tall=$tall: $(tail -1 gpurun_out/bench256_tall$tall.log | cut -c1-160)"
done
done
exit 0
