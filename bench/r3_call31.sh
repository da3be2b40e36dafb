#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out/r3 && export PYTHONPATH=$R
timeout -k 10 900 python -u -m pytest tests/test_gpu.py -m gpu -x -q --timeout 300 --timeout-method thread \
  -k "gemm or gemm_core or epilogue_takes or conv_kernels or dataflow or vit or toy" > gpurun_out/r3/stage_tests.log 2>&1
rc=$?; tail -4 gpurun_out/r3/stage_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -u bench/micro/gemm_check.py 2>&1 | grep -E "M=25216|M=25088|M=8192|M=4096|FAIL|ALL" || exit 1
timeout -k 10 300 python -u bench/micro/gemm_breakdown_vit.py 512 2>&1 | grep -E "^N=" | cut -c1-60 || exit 1
echo "== vit B=512"
timeout -k 10 300 python -u bench.py --model vit_b_16 --optimizer adamw --batch_size 512 --steps 15 --warmup 4 2>/dev/null | cut -c1-180 || exit 1
echo "== resnet50 default"
timeout -k 10 400 python -u bench.py --steps 12 --warmup 4 2>/dev/null | cut -c1-180 || exit 1
echo "== resnet50 b1280"
timeout -k 10 400 python -u bench.py --steps 20 --warmup 5 --batch_size 1280 2>/dev/null | cut -c1-180 || exit 1
