#!/bin/bash
# round 3, GPU call 1: suite on the new tree, N=1 with / without the size-1 communicator, --pmc passes at B=1280
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out/r3 && export PYTHONPATH=$R
step() { local name=$1 secs=$2; shift 2
  timeout -k 10 $secs "$@" > gpurun_out/r3/$name.log 2>&1; local rc=$?
  echo "[$name] rc=$rc: $(tail -1 gpurun_out/r3/$name.log | cut -c1-300)"
  if [ $rc -ne 0 ]; then tail -40 gpurun_out/r3/$name.log; exit $rc; fi; }
step pytest_gpu 900 python -u -m pytest tests -m gpu -q -x --timeout 300 --timeout-method thread
step bench_comm1 300 python -u bench.py --steps 10 --warmup 4
PDT_FORCE_COMM=0 step bench_comm0 300 python -u bench.py --steps 10 --warmup 4
step bench_b1280 300 python -u bench.py --steps 10 --warmup 4 --batch_size 1280
PMC_BATCH=1280 bash bench/pmc_profile.sh
