#!/bin/bash
# One gpurun call that validates a tree on an MI355X: GPU test suite, smoke, kernel numerics + timing,
# the headline bench and a rocprofv3 kernel-stats profile.  Every GPU step has its own time limit and the
# script stops at the first step that fails (no retries on a faulted GPU).
#   /usr/local/graft/bin/gpurun --timeout 1190 -- 'bash bench/gpu_ci.sh'
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out && export PYTHONPATH=$R
step() { # name, seconds, command...
  local name=$1 secs=$2; shift 2
  timeout -k 10 $secs "$@" > gpurun_out/$name.log 2>&1
  local rc=$?
  echo "[$name] rc=$rc: $(tail -1 gpurun_out/$name.log | cut -c1-240)"
  if [ $rc -ne 0 ]; then tail -30 gpurun_out/$name.log; exit $rc; fi
}
step pytest_gpu 900 python -u -m pytest tests -m gpu -q -x --timeout 300 --timeout-method thread
step smoke 300 python -u -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
step kernel_check 500 python -u bench/kernel_check.py --time
step bench_default 300 python -u bench.py --steps 10 --warmup 4
step bench_b1024 300 python -u bench.py --steps 10 --warmup 4 --batch_size 1024
step bench_b256 300 python -u bench.py --steps 15 --warmup 5 --batch_size 256
step bench_cifar_eager 300 python -u bench.py --image_size 32 --optimizer adam --sync_bn --batch_size 128 --steps 60 --warmup 15
step bench_cifar_graph 300 python -u bench.py --image_size 32 --optimizer adam --sync_bn --batch_size 128 --steps 60 --warmup 15 --graph
step bench_torchrun 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 8 --warmup 3
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_default -- \
  python3 $R/bench.py --steps 3 --warmup 2 > $R/gpurun_out/prof_default.log 2>&1
echo "[rocprof] rc=$?"
rm -f $R/gpurun_out/prof_default/*/*kernel_trace.csv   # the per-dispatch trace is large; the stats CSV is what gets kept
exit 0
