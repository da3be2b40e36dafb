#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout -k 10 600 python -u -m pytest tests/test_gpu.py -m gpu -q -x --timeout 300 --timeout-method thread -k "graph or size1 or stem" > gpurun_out/pytest_gpu_new.log 2>&1
rc=$?; echo "pytest new rc=$rc"; grep -E "passed|failed|^FAILED|Error|capture" gpurun_out/pytest_gpu_new.log | tail -10
if [ $rc -ne 0 ]; then tail -40 gpurun_out/pytest_gpu_new.log; exit $rc; fi
run() { local name=$1; shift
  timeout -k 10 300 python -u bench.py "$@" > gpurun_out/bench_$name.log 2>&1
  local rc=$?; echo "$name rc=$rc: $(tail -1 gpurun_out/bench_$name.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"].get("hip_graph"))' 2>/dev/null || tail -3 gpurun_out/bench_$name.log)"
}
run cifar_b128_eager --steps 40 --warmup 10 --batch_size 128 --image_size 32 --optimizer adam --sync_bn
run cifar_b128_graph --steps 40 --warmup 10 --batch_size 128 --image_size 32 --optimizer adam --sync_bn --graph
run cifar_b1024_eager --steps 20 --warmup 6 --batch_size 1024 --image_size 32 --optimizer adam --sync_bn
run cifar_b1024_graph --steps 20 --warmup 6 --batch_size 1024 --image_size 32 --optimizer adam --sync_bn --graph
run cifar_b4096_eager --steps 10 --warmup 5 --batch_size 4096 --image_size 32 --optimizer adam --sync_bn
run r50_b256_graph --steps 15 --warmup 6 --batch_size 256 --graph
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_vit -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --batch_size 128 --model vit_b_16 --optimizer adamw > $GRAFT_REPO_ROOT/gpurun_out/prof_vit.log 2>&1
echo "vit prof rc=$?"
rm -f $GRAFT_REPO_ROOT/gpurun_out/prof_vit/*/*kernel_trace.csv
exit 0
