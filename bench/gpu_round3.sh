#!/bin/bash
# Re-check kernels after the partial-stats / kernarg-flatten rework; bench; big-shape kernel timings.
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout -k 10 300 python -u bench/kernel_check.py --quick --out gpurun_out/kc_quick.json > gpurun_out/kc_quick.log 2>&1
rc=$?; echo "kernel_check quick rc=$rc"; grep -E "^FAIL|cases" gpurun_out/kc_quick.log | cut -c1-250
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then tail -20 gpurun_out/kc_quick.log; exit $rc; fi
timeout -k 10 300 python -u bench.py --steps 10 --warmup 3 --batch_size 128 > gpurun_out/bench_native_b128.log 2>&1
rc=$?; echo "native bench b128 rc=$rc"; tail -1 gpurun_out/bench_native_b128.log | cut -c1-300
if [ $rc -ne 0 ]; then tail -30 gpurun_out/bench_native_b128.log; exit $rc; fi
timeout -k 10 300 python -u bench.py --steps 15 --warmup 5 --batch_size 256 > gpurun_out/bench_native_b256.log 2>&1
rc=$?; echo "native bench b256 rc=$rc"; tail -1 gpurun_out/bench_native_b256.log | cut -c1-300
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 400 python -u bench/kernel_check.py --time --only bn,conv --out gpurun_out/kc_full.json > gpurun_out/kc_full.log 2>&1
rc=$?; echo "kernel_check full rc=$rc"; grep -E "^FAIL|cases" gpurun_out/kc_full.log | cut -c1-250
exit 0
