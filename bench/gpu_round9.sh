#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH=$PWD
run() { local name=$1; shift
  timeout -k 10 300 python -u bench.py "$@" > gpurun_out/bench_$name.log 2>&1
  local rc=$?; echo "$name rc=$rc: $(tail -1 gpurun_out/bench_$name.log | cut -c1-220)"
  if [ $rc -ne 0 ]; then tail -20 gpurun_out/bench_$name.log; exit $rc; fi
}
run r50_b768 --steps 8 --warmup 3 --batch_size 768
run r50_b1024 --steps 8 --warmup 3 --batch_size 1024
bash bench/pmc_profile.sh
