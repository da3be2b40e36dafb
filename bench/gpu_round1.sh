#!/bin/bash
# First GPU pass: kernel numerics/timing sweep, stock-PyTorch calibration, native bench.
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout -k 10 400 python -u bench/kernel_check.py --quick --time --out gpurun_out/kc_quick.json > gpurun_out/kc_quick.log 2>&1
rc=$?; echo "kernel_check quick rc=$rc"; tail -5 gpurun_out/kc_quick.log
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
timeout -k 10 300 python -u bench/kernel_check.py --time --only conv --out gpurun_out/kc_conv.json > gpurun_out/kc_conv.log 2>&1
rc=$?; echo "kernel_check conv rc=$rc"; tail -3 gpurun_out/kc_conv.log
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
PDT_PLAIN_TORCH=1 PDT_FORCE_REFERENCE_OPS=1 timeout -k 10 300 python -u bench.py --steps 10 --warmup 3 --batch_size 128 > gpurun_out/bench_plain_torch.log 2>&1
rc=$?; echo "plain torch bench rc=$rc"; tail -2 gpurun_out/bench_plain_torch.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python -u bench.py --steps 10 --warmup 3 --batch_size 128 > gpurun_out/bench_native.log 2>&1
rc=$?; echo "native bench rc=$rc"; tail -2 gpurun_out/bench_native.log
exit $rc
