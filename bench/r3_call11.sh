#!/bin/bash
# W=2 rehearsal, SyncBN exchange inside the BN kernels AND buckets through the two-shot IPC kernel on the reducer's own stream:
# first with 16 hardware queues per process, then (only if that completes) SyncBN through the group with 4 queues
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out/r3 && export PYTHONPATH=$R
export PDT_TRACE_COLL=1 DEADLINE=40 PDT_GRAD_COMM=ipc PDT_IPC_TIMEOUT_S=20
GPU_MAX_HW_QUEUES=16 PDT_SYNCBN_COMM=ipc TAG=ii_q16_1 PORT=29661 bash bench/micro/w2_one_gpu.sh && \
GPU_MAX_HW_QUEUES=16 PDT_SYNCBN_COMM=ipc TAG=ii_q16_2 PORT=29662 bash bench/micro/w2_one_gpu.sh && \
PDT_SYNCBN_COMM=auto TAG=ai_q4_1 PORT=29663 bash bench/micro/w2_one_gpu.sh && \
PDT_SYNCBN_COMM=auto TAG=ai_q4_2 PORT=29664 bash bench/micro/w2_one_gpu.sh
echo "w2 rc=$?"
timeout -k 10 300 python -u bench/micro/eager_host_profile.py > gpurun_out/r3/eager_host_profile.log 2>&1 || { tail -20 gpurun_out/r3/eager_host_profile.log; exit 1; }
head -50 gpurun_out/r3/eager_host_profile.log
