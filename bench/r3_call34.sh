#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out/r3 && export PYTHONPATH=$R
run() { # tag, env...
  local tag=$1; shift
  env "$@" timeout -k 10 200 python -u bench/convergence.py --model vit_b_16 --image_size 224 --num_classes 10 --batch_size 64 --steps 150 --window 50 \
    --optimizer adamw --lr 3e-4 --noise 0.1 --backend native 2>/dev/null | grep '^{' | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('$tag', d['loss_curve'], d['eval_top1'])" || echo "$tag FAILED"
}
run default A=1
run attn_old PDT_ATTN_PIPE=0
run attn_generic PDT_ATTN_GENERIC=1
