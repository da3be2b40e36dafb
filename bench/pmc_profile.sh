#!/bin/bash
# rocprofv3 counter passes over a short bench.py run (each pass its own process; counters only).
#   PMC_BATCH=256 PMC_EXTRA="--model vit_b_16 --optimizer adamw" PMC_OUT=pmc_vit bash bench/pmc_profile.sh
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
export PYTHONPATH=$R
OUT=${PMC_OUT:-pmc}
mkdir -p $R/gpurun_out/$OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 2 --batch_size ${PMC_BATCH:-256} ${PMC_EXTRA:-}"
pass() { # name counters...
  local name=$1; shift
  timeout -s KILL 280 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/$OUT/$name -- python3 $R/bench.py $ARGS > $R/gpurun_out/$OUT/$name.log 2>&1
  local rc=$?; echo "pmc pass $name rc=$rc"; [ $rc -eq 0 ] || { tail -5 $R/gpurun_out/$OUT/$name.log; exit $rc; }
}
pass sq SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
pass fetch FETCH_SIZE
pass write WRITE_SIZE
cd $R && python3 bench/pmc_summarize.py gpurun_out/$OUT > gpurun_out/$OUT/summary.md 2>gpurun_out/$OUT/summary.err
echo "summary rc=$?"; head -30 gpurun_out/$OUT/summary.md
# the raw per-dispatch CSVs are large; keep only the summary + logs in the merge
find gpurun_out/$OUT -name "*counter_collection.csv" -size +20M -delete
exit 0
