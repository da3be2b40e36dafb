#!/bin/bash
# stall-reason counter passes over bench/micro/conv_one.py (each pass its own process; counters only)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
export PYTHONPATH=$R
mkdir -p $R/gpurun_out/convpmc
cd /tmp && export TMPDIR=/tmp
pass() {
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/convpmc/$name -- python3 $R/bench/micro/conv_one.py ${CONV_ARGS} > $R/gpurun_out/convpmc/$name.log 2>&1
  local rc=$?; echo "pmc pass $name rc=$rc"; [ $rc -eq 0 ] || { tail -5 $R/gpurun_out/convpmc/$name.log; exit $rc; }
}
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
pass b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM
pass c SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_MFMA SQ_WAVES
exit 0
