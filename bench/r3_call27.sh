#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out/r3 && export PYTHONPATH=$R
timeout -k 10 600 python -u -m pytest tests/test_gpu.py -m gpu -x -q --timeout 200 --timeout-method thread -k "wgrad or chunk or window" 2>&1 | tail -4 || exit 1
echo "== vit default batch"
timeout -k 10 400 python -u bench.py --model vit_b_16 --optimizer adamw --steps 8 --warmup 3 2>/dev/null | cut -c1-180 || exit 1
echo "== resnet50 default"
timeout -k 10 400 python -u bench.py --steps 12 --warmup 4 2>/dev/null | cut -c1-180 || exit 1
