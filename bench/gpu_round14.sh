#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout -k 10 200 python -u bench/kernel_check.py --time --only attn > gpurun_out/kc_attn.log 2>&1
rc=$?; echo "attn check rc=$rc"; grep -E "attention|Error|error" gpurun_out/kc_attn.log | cut -c1-230
if [ $rc -gt 1 ]; then tail -20 gpurun_out/kc_attn.log; exit $rc; fi
if [ $rc -eq 1 ]; then exit 1; fi
timeout -k 10 300 python -u bench.py --steps 10 --warmup 6 --batch_size 128 --model vit_b_16 --optimizer adamw > gpurun_out/bench_vit.log 2>&1
echo "vit rc=$?: $(tail -1 gpurun_out/bench_vit.log | cut -c1-250)"
timeout -k 10 300 python -u bench.py --steps 10 --warmup 6 --batch_size 256 --model vit_b_16 --optimizer adamw > gpurun_out/bench_vit256.log 2>&1
echo "vit256 rc=$?: $(tail -1 gpurun_out/bench_vit256.log | cut -c1-250)"
timeout -k 10 600 python -u -m pytest tests/test_gpu.py -m gpu -q -x --timeout 300 --timeout-method thread -k "attention or vit" > gpurun_out/pytest_gpu_new.log 2>&1
rc=$?; echo "pytest new rc=$rc"; grep -E "passed|failed|^FAILED|Error" gpurun_out/pytest_gpu_new.log | tail -10
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_vit -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 6 --batch_size 128 --model vit_b_16 --optimizer adamw > $GRAFT_REPO_ROOT/gpurun_out/prof_vit.log 2>&1
echo "vit prof rc=$?"
rm -f $GRAFT_REPO_ROOT/gpurun_out/prof_vit/*/*kernel_trace.csv
exit 0
