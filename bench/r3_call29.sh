#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out/r3
hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/gemm_v2_probe bench/micro/gemm_v2_probe.hip 2>/dev/null || exit 1
for shape in "8192 8192 8192" "4096 4096 4096" "25088 768 768" "25088 2304 768" "25088 3072 768" "25088 768 3072" "100352 768 768" "100352 3072 768"; do
  timeout -k 10 60 /tmp/gemm_v2_probe $shape 0 || exit 1
  timeout -k 10 60 /tmp/gemm_v2_probe $shape 1 || exit 1
done 2>&1 | tee gpurun_out/r3/gemm_v2_probe.log
