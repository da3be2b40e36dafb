#!/bin/bash
# 128-wide tile build of the GEMM core: tests, then timing of the ViT shapes at each width, then step-level A/B
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out/r3 && export PYTHONPATH=$R
timeout -k 10 900 python -u -m pytest tests/test_gpu.py -m gpu -x -q --timeout 300 --timeout-method thread \
  -k "gemm or gemm_core or epilogue_takes" > gpurun_out/r3/bnt_tests.log 2>&1
rc=$?; tail -5 gpurun_out/r3/bnt_tests.log
[ $rc -eq 0 ] || exit $rc
for bn in 256 128 0; do
  echo "== gemm_check --bn $bn"
  timeout -k 10 300 python -u bench/micro/gemm_check.py --bn $bn 2>&1 | grep -E "M=25216|M=25088|FAIL|ALL" || exit 1
done > gpurun_out/r3/bnt_gemm_shapes.log 2>&1
cat gpurun_out/r3/bnt_gemm_shapes.log
echo "== vit auto"
timeout -k 10 300 python -u bench.py --model vit_b_16 --optimizer adamw --batch_size 128 --steps 20 --warmup 5 2>/dev/null | cut -c1-160 || exit 1
echo "== vit 256"
PDT_GEMM_BN=256 timeout -k 10 300 python -u bench.py --model vit_b_16 --optimizer adamw --batch_size 128 --steps 20 --warmup 5 2>/dev/null | cut -c1-160 || exit 1
echo "== resnet default"
timeout -k 10 300 python -u bench.py --batch_size 1280 --steps 20 --warmup 5 2>/dev/null | cut -c1-160 || exit 1
echo "== resnet N128 on the core"
PDT_CONV_GEMM_N128=1 timeout -k 10 300 python -u bench.py --batch_size 1280 --steps 20 --warmup 5 2>/dev/null | cut -c1-160 || exit 1
