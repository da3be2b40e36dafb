#!/bin/bash
# round 3, GPU call 2: GEMM-core time breakdown, then the W=2-on-one-GPU rehearsal with collective tracing
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out/r3 && export PYTHONPATH=$R
timeout -k 10 300 python -u bench/micro/gemm_breakdown.py 1280 > gpurun_out/r3/gemm_breakdown.log 2>&1 || { tail -20 gpurun_out/r3/gemm_breakdown.log; exit 1; }
cat gpurun_out/r3/gemm_breakdown.log
export PDT_TRACE_COLL=1 DEADLINE=60
TAG=a1 PORT=29571 bash bench/micro/w2_one_gpu.sh && \
TAG=a2 PORT=29572 bash bench/micro/w2_one_gpu.sh && \
TAG=a3 PORT=29573 bash bench/micro/w2_one_gpu.sh && \
TAG=py PORT=29574 PDT_REDUCER=python bash bench/micro/w2_one_gpu.sh
echo "w2 rc=$?"
