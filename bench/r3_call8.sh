#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out/r3 && export PYTHONPATH=$R
A="--model resnet18 --image_size 64 --num_classes 100 --batch_size 64 --steps 80 --window 10 --eval_batches 8 --noise 0.4"
rm -f gpurun_out/r3/conv_short2.jsonl
for seed in 0 1 2; do
  for be in native oracle; do
    envs="X=1"; [ $be = oracle ] && envs="PDT_FORCE_REFERENCE_OPS=1"
    env $envs timeout -k 10 200 python -u bench/convergence.py --backend $be --seed $seed $A 2>/dev/null | tail -1 >> gpurun_out/r3/conv_short2.jsonl || exit 1
  done
done
timeout -k 10 200 python -u bench/convergence.py --backend native --seed 0 --graph 1 --grad_dtype bf16 $A 2>/dev/null | tail -1 >> gpurun_out/r3/conv_short2.jsonl || exit 1
python - <<'PY'
import json
for l in open("gpurun_out/r3/conv_short2.jsonl"):
    d=json.loads(l); print(d["backend"], d["seed"], d["graph"], d["loss_curve"], d["eval_top1"])
PY
timeout -k 10 400 python -u -m pytest tests/test_gpu.py -m gpu -x -q --timeout 300 --timeout-method thread -k "follows_the_fp32" 2>&1 | tail -15
