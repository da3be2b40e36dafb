# Same-box sweep of the tuning switches against the defaults: bash bench/micro/knob_sweep.sh [bench args]
run() { env "$@" timeout -k 10 200 python bench.py --steps 10 --warmup 4 $ARGS 2>/dev/null | sed 's/.*"value": \([0-9.]*\).*"ms_per_step": \([0-9.]*\).*/\1 img\/s \2 ms/' | sed "s/^/$* : /"; }
ARGS="$@"
run PDT_NONE=1
for kv in PDT_FUSED_DGRAD_IGEMM=0 PDT_IGEMM_XCD_GROUP=16 PDT_WGRAD_TN_BLOCKS=512 PDT_WGRAD_TN_BLOCKS=128 PDT_WGRAD_BLOCKS=512 PDT_WGRAD_BLOCKS=1024 PDT_GEMM_GRID=512 PDT_CONV_GEMM3=0 PDT_IGEMM_TALL=0 PDT_WGRAD_TN_MIN_M=1000000000 PDT_CONV_WIN64=0 PDT_IGEMM_BNR_FAT=1 PDT_BN_GRID_CAP=4096; do
  run $kv || exit 1
done
run PDT_NONE=1
