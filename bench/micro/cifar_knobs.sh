#!/bin/bash
# Same-box sweep of the data-flow switches on the launch-bound CIFAR-shaped step (hipGraph replay): each switch off
# in turn, baseline before and after.   /usr/local/graft/bin/gpurun --timeout 600 -- 'bash bench/micro/cifar_knobs.sh'
R=${GRAFT_REPO_ROOT:-$PWD}
ARGS="--steps 300 --warmup 30 --batch_size 128 --image_size 32 --sync_bn --optimizer adam --graph"
one() { # label, env assignment (or nothing)
  local label=$1; shift
  local out
  out=$(env "$@" timeout -k 10 120 python $R/bench.py $ARGS 2>/dev/null) || { echo "$label: failed"; exit 1; }
  echo "$label: $(echo "$out" | sed 's/.*"value": \([0-9.]*\).*"ms_per_step": \([0-9.]*\).*/\1 img\/s \2 ms/')"
}
one baseline PDT_NOOP=1
for k in PDT_FUSE_STEM_POOL_BWD PDT_DUAL_BN_BACKWARD PDT_POOL_BWD_2X2 PDT_FUSE_STEM_POOL PDT_FUSE_BN_REDUCE PDT_LAZY_RES_GRAD PDT_DEFER_DS_BN PDT_COMPACT_DS_GRAD PDT_CONV_WIN64 PDT_WGRAD_WIN64 PDT_STEM_WIN; do
  one "$k=0" $k=0
done
one baseline PDT_NOOP=1
