#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out/r3 && export PYTHONPATH=$R
for s in 0 1; do
timeout -k 10 500 python -u bench/convergence.py --model vit_b_16 --image_size 224 --num_classes 10 --batch_size 64 --steps 300 --window 50 \
  --optimizer adamw --lr 1e-4 --noise 0.1 --backends native,oracle --seed $s --timeout 400 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['backend'], 'seed', d['seed'], d['loss_curve'], d['eval_top1'], d['finite'])" || exit 1
done
