#!/bin/bash
# round 3, GPU call 7: calibration of the short convergence test (ResNet-18, 64x64, 60 steps), then the new W=2 tests
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out/r3 && export PYTHONPATH=$R
A="--model resnet18 --image_size 64 --num_classes 20 --batch_size 64 --steps 60 --window 15 --eval_batches 8"
for seed in 0 1 2; do
  for be in native oracle; do
    envs=""; [ $be = oracle ] && envs="PDT_FORCE_REFERENCE_OPS=1"
    env $envs timeout -k 10 200 python -u bench/convergence.py --backend $be --seed $seed $A 2>/dev/null | tail -1 >> gpurun_out/r3/conv_short.jsonl || exit 1
  done
  timeout -k 10 200 python -u bench/convergence.py --backend native --seed $seed --graph 1 --grad_dtype bf16 $A 2>/dev/null | tail -1 >> gpurun_out/r3/conv_short.jsonl || exit 1
done
cat gpurun_out/r3/conv_short.jsonl | cut -c1-400
timeout -k 10 400 python -u -m pytest tests/test_gpu.py -m gpu -x -q --timeout 300 --timeout-method thread -k "two_engine_ranks" 2>&1 | tail -5
