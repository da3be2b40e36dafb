#!/bin/bash
# round 3, GPU call 5: new attention / LayerNorm kernels, then the W=2 rehearsal with host-staged gloo collectives
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out/r3 && export PYTHONPATH=$R
timeout -k 10 600 python -u -m pytest tests/test_gpu.py -m gpu -x -q --timeout 300 --timeout-method thread -k "attention or layernorm or vit or fallback" > gpurun_out/r3/pytest_attn.log 2>&1
rc=$?; tail -15 gpurun_out/r3/pytest_attn.log; [ $rc -eq 0 ] || exit $rc
export PDT_TRACE_COLL=1 DEADLINE=45
TAG=hs1 PORT=29601 bash bench/micro/w2_one_gpu.sh && \
TAG=hs2 PORT=29602 bash bench/micro/w2_one_gpu.sh && \
TAG=hs3 PORT=29603 bash bench/micro/w2_one_gpu.sh && \
TAG=hs4 PORT=29604 bash bench/micro/w2_one_gpu.sh
echo "w2 rc=$?"
