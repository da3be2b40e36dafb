#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH=$PWD
R=$PWD
run() { # name, args...
  local name=$1; shift
  timeout -k 10 300 python -u bench.py "$@" > gpurun_out/bench_$name.log 2>&1
  local rc=$?; echo "$name rc=$rc: $(tail -1 gpurun_out/bench_$name.log | cut -c1-220)"
  if [ $rc -ne 0 ]; then tail -20 gpurun_out/bench_$name.log; exit $rc; fi
}
run r50_b512 --steps 10 --warmup 4 --batch_size 512
run r50_b128 --steps 20 --warmup 5 --batch_size 128
run r50_b512_syncbn --steps 10 --warmup 4 --batch_size 512 --sync_bn
run r152_b128 --steps 10 --warmup 4 --batch_size 128 --model resnet152
run r18_b512 --steps 10 --warmup 4 --batch_size 512 --model resnet18
run vit_b128 --steps 10 --warmup 4 --batch_size 128 --model vit_b_16 --optimizer adamw
run r50_cifar_b128 --steps 30 --warmup 10 --batch_size 128 --image_size 32 --optimizer adam --sync_bn
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --list-avail > $R/gpurun_out/rocprof_list_avail.txt 2>&1
echo "list-avail rc=$?"
exit 0
