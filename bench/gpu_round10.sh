#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout -k 10 300 python -u bench/kernel_check.py --quick --only stem,conv > gpurun_out/kc_quick.log 2>&1
rc=$?; echo "kernel_check quick rc=$rc"; grep -E "FAIL|cases," gpurun_out/kc_quick.log | head -20
if [ $rc -ne 0 ]; then tail -5 gpurun_out/kc_quick.log; exit $rc; fi
timeout -k 10 300 python -u bench/kernel_check.py --time --only stem > gpurun_out/kc_stem.log 2>&1
echo "stem timing rc=$?"; grep -E "stem_" gpurun_out/kc_stem.log | cut -c1-200
timeout -k 10 300 python -u bench/resnet50_roofline.py --batch 256 > gpurun_out/roofline_b256_win.md 2>&1
echo "roofline rc=$?"; grep -E "conv_wgrad.*k3" gpurun_out/roofline_b256_win.md; tail -7 gpurun_out/roofline_b256_win.md
PDT_WGRAD_WIN=0 timeout -k 10 300 python -u bench/resnet50_roofline.py --batch 256 --out gpurun_out/roofline_old.json > gpurun_out/roofline_b256_old.md 2>&1
echo "old:"; grep -E "conv_wgrad.*k3" gpurun_out/roofline_b256_old.md
timeout -k 10 300 python -u bench.py --steps 10 --warmup 4 --batch_size 256 > gpurun_out/bench_b256.log 2>&1
rc=$?; echo "b256 rc=$rc: $(tail -1 gpurun_out/bench_b256.log | cut -c1-200)"
[ $rc -eq 0 ] || { tail -20 gpurun_out/bench_b256.log; exit $rc; }
timeout -k 10 300 python -u bench.py --steps 8 --warmup 3 > gpurun_out/bench_b1024.log 2>&1
rc=$?; echo "b1024 rc=$rc: $(tail -1 gpurun_out/bench_b1024.log | cut -c1-200)"
exit $rc
