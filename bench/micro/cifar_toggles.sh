cd $GRAFT_REPO_ROOT
export PDT_QUIET=1
ARGS="--steps 30 --warmup 10 --batch_size 128 --image_size 32 --num_classes 10 --sync_bn --optimizer adam --graph"
for t in NONE PDT_CONV_GEMM PDT_CONV_GEMM3 PDT_WGRAD_TN PDT_CONV_WIN64 ALL; do
  if [ $t = ALL ]; then export PDT_CONV_GEMM=0 PDT_CONV_GEMM3=0 PDT_WGRAD_TN=0 PDT_CONV_WIN64=0; v="env"; elif [ $t = NONE ]; then v="env"; else v="env $t=0"; fi
  $v timeout -k 10 200 python bench.py $ARGS > gpurun_out/cifar_t_$t.log 2>&1
  echo "$t off: $(tail -n 1 gpurun_out/cifar_t_$t.log | sed 's/.*"value": \([0-9.]*\).*"ms_per_step": \([0-9.]*\).*/\1 img\/s \2 ms/')"
done
