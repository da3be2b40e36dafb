#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout -k 10 400 python -u bench/kernel_check.py --quick --only conv > gpurun_out/kc_conv.log 2>&1
rc=$?; echo "conv check rc=$rc"; grep -E "FAIL|cases," gpurun_out/kc_conv.log | cut -c1-200
if [ $rc -ne 0 ]; then tail -20 gpurun_out/kc_conv.log; exit $rc; fi
timeout -k 10 300 python -u bench/resnet50_roofline.py --batch 256 > gpurun_out/roofline_b256_x.md 2>&1
echo "roofline rc=$?"; grep -E "conv_wgrad" gpurun_out/roofline_b256_x.md | head -30; tail -7 gpurun_out/roofline_b256_x.md
timeout -k 10 300 python -u bench.py --steps 10 --warmup 4 > gpurun_out/bench_b1024.log 2>&1
echo "b1024 rc=$?: $(tail -1 gpurun_out/bench_b1024.log | cut -c1-200)"
exit 0
