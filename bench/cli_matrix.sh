#!/bin/bash
# main.py (the reference-compatible CLI) through a matrix of flag combinations on one GPU: every run must exit 0,
# write its summary and end with a finite loss below the first one.  One line per combination.
#   /usr/local/graft/bin/gpurun --timeout 1100 -- 'bash bench/cli_matrix.sh'
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/cli_matrix
rm -rf $OUT && mkdir -p $OUT && cd $OUT
export PYTHONPATH=$R MASTER_ADDR=127.0.0.1
port=29600
MODE=
fail=0
run() { # name, launcher (direct|launch|torchrun), main.py args...
  local name=$1 how=$2; shift 2
  port=$((port + 1))
  local common="--dataset synthetic-cifar --dataset_len 3840 --num_classes 100 --epochs 1 --steps 30 --summary_json $OUT/$name.json --log_dir $OUT/$name"
  case $how in
    direct)   MASTER_PORT=$port timeout -k 10 240 python -u $R/main.py $common "$@" > $name.log 2>&1 ;;
    launch)   timeout -k 10 240 python -u -m pytorch_distributed_training_amd.launch --nproc_per_node 1 --master_addr 127.0.0.1 --master_port $port $R/main.py $common "$@" > $name.log 2>&1 ;;
    torchrun) timeout -k 10 240 python -u -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $port $R/main.py $common "$@" > $name.log 2>&1 ;;
  esac
  local rc=$?
  local verdict
  verdict=$(python $R/bench/cli_matrix_verdict.py $OUT/$name.json $OUT/$name.log $MODE)
  echo "[$name] rc=$rc $verdict"
  case $rc in 124|134|137|139) grep -v 'ProcessGroupNCCL\|^$' $name.log | tail -6 | cut -c1-300; echo "stopping: no further GPU step after rc=$rc"; exit $rc ;; esac
  if [ $rc -ne 0 ]; then grep -v 'ProcessGroupNCCL\|^$' $name.log | tail -4 | cut -c1-300; fail=1; fi
  case "$verdict" in OK*) ;; *) fail=1 ;; esac
}
run default            direct
run no_graph           direct   --no_graph
run bf16_buckets       direct   --grad_dtype bf16
run no_sync_bn_sgd     direct   --no_sync_bn --optimizer sgd --lr 0.01
run fp32               direct   --dtype fp32 --no_graph
run resnet18_eval      direct   --model resnet18 --eval
run vit_tiny           direct   --model vit_tiny_test --optimizer adamw
run profile_none       direct   --profile none
run profile_roctx      direct   --profile roctx
run debug_replicas     direct   --debug_replicas --no_graph
run bucket_auto        direct   --bucket_cap_mb auto
run own_launcher       launch
run torchrun           torchrun
run checkpoint         direct   --checkpoint $OUT/ck.pt --checkpoint_every 10
run resume             direct   --resume $OUT/ck.pt --epochs 2 --steps 45
MODE=finite-only  # 24 steps over 1000 classes: Adam on a fresh BN network has not turned the loss around yet
run imagenet_shape     direct   --dataset synthetic-imagenet --num_classes 1000 --dataset_len 1024 --batch_size 32 --steps 24 --lr 0.0003
# a head narrower than the label range is refused up front (the loss kernel would only turn the loss into NaN)
MASTER_PORT=$((port + 1)) timeout -k 10 120 python -u $R/main.py --dataset synthetic-imagenet --num_classes 100 --dataset_len 256 --batch_size 32 --steps 2 > narrow_head.log 2>&1
rc=$?
if [ $rc -ne 0 ] && grep -q "narrower than" narrow_head.log; then echo "[narrow_head_refused] OK rc=$rc"; else echo "[narrow_head_refused] CHECK rc=$rc"; fail=1; fi
# large artefacts (checkpoints, profiler traces) stay on the box: gpurun_out is capped at 64 MiB
find $OUT -size +2M -delete
exit $fail
