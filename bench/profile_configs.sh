#!/bin/bash
# bench JSON + rocprofv3 kernel stats for the secondary BASELINE.json configs (ViT-B/16, ResNet-152, SyncBN large
# batch, CIFAR shape); results under gpurun_out/cfg_*.  One process per step, each with its own time limit.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
export PYTHONPATH=$R
mkdir -p $R/gpurun_out
run() { # name, bench args...
  local name=$1; shift
  timeout -k 10 300 python -u $R/bench.py "$@" > $R/gpurun_out/cfg_$name.log 2>&1 || { echo "[$name] bench failed"; tail -5 $R/gpurun_out/cfg_$name.log; exit 1; }
  echo "[$name] $(tail -1 $R/gpurun_out/cfg_$name.log | cut -c1-200)"
  (cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/cfg_prof_$name -- \
     python3 $R/bench.py "$@" --steps 3 --warmup 3 > $R/gpurun_out/cfg_prof_$name.log 2>&1) || { echo "[$name] rocprof failed"; exit 1; }
  rm -f $R/gpurun_out/cfg_prof_$name/*/*kernel_trace.csv
}
run vit_b16 --model vit_b_16 --optimizer adamw --batch_size 128 --steps 20 --warmup 5
run resnet152 --model resnet152 --batch_size 256 --steps 10 --warmup 4
run resnet50_syncbn_b1024 --sync_bn --batch_size 1024 --steps 8 --warmup 3
run cifar_graph --image_size 32 --optimizer adam --sync_bn --batch_size 128 --steps 60 --warmup 15 --graph
exit 0
