#!/bin/bash
# per-shape A/B of implicit-GEMM tile configurations (PDT_IGEMM_CFG) with event timing
R=${GRAFT_REPO_ROOT:-$PWD}
export PYTHONPATH=$R
SHAPES=${SHAPES:-"256,14,1024,1,1 1024,14,256,1,1 128,28,512,1,1 512,28,128,1,1 512,7,2048,1,1 2048,7,512,1,1"}
for cfg in ${CFGS:-0 10 12 8}; do
  PDT_IGEMM_CFG=$cfg timeout -k 10 120 python -u $R/bench/micro/conv_one.py --time --batch ${BATCH:-1024} --shapes $SHAPES 2>&1 | grep "^time" || exit 1
done
