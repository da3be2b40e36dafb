#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out/r3 && export PYTHONPATH=$R
run() { # tag, seed, env...
  local tag=$1 seed=$2; shift 2
  env "$@" timeout -k 10 200 python -u bench/convergence.py --model vit_b_16 --image_size 224 --num_classes 10 --batch_size 64 --steps 150 --window 50 \
    --optimizer adamw --lr 3e-4 --noise 0.1 --backend native --seed $seed 2>/dev/null | grep '^{' | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('$tag seed $seed', d['loss_curve'], d['eval_top1'])" || echo "$tag FAILED"
}
for s in 1 2 3; do
run pipelined $s A=1
run generic $s PDT_ATTN_GENERIC=1
done
run pipelined 0 A=1
