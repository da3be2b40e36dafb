#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH=$PWD
for cfg in 0 1 2 5; do
PDT_IGEMM_CFG=$cfg timeout -k 10 300 python -u bench.py --steps 10 --warmup 4 > gpurun_out/bench_cfg$cfg.log 2>&1
echo "b1024 cfg=$cfg: $(tail -1 gpurun_out/bench_cfg$cfg.log | cut -c1-160)"
done
exit 0
