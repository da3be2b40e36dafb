#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout -k 10 300 python -u bench/resnet50_roofline.py --batch 256 > gpurun_out/roofline_b256.md 2>&1
rc=$?; echo "roofline rc=$rc"; tail -8 gpurun_out/roofline_b256.md
if [ $rc -ne 0 ]; then tail -30 gpurun_out/roofline_b256.md; exit $rc; fi
timeout -k 10 300 python -u bench.py --steps 15 --warmup 5 --batch_size 256 > gpurun_out/bench_native_b256.log 2>&1
rc=$?; echo "native bench b256 rc=$rc"; tail -1 gpurun_out/bench_native_b256.log | cut -c1-300
exit $rc
