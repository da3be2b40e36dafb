#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out/r3 && export PYTHONPATH=$R
timeout -k 10 600 python -u -m pytest tests/test_gpu.py -m gpu -x -q --timeout 200 --timeout-method thread -k "attention" 2>&1 | tail -4 || exit 1
timeout -k 10 400 python -u bench/micro/vit_nan_probe.py 2>&1 | grep -E "^step|^   " | tail -12
timeout -k 10 500 python -u bench/convergence.py --model vit_b_16 --image_size 224 --num_classes 10 --batch_size 64 --steps 400 --window 50 \
  --optimizer adamw --lr 3e-4 --noise 0.1 --backend native 2>/dev/null | grep '^{' | cut -c1-400
