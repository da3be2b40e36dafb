cd $GRAFT_REPO_ROOT
export PDT_QUIET=1
run() { name=$1; shift; timeout -k 10 280 python bench.py "$@" > gpurun_out/cfg_$name.log 2>&1; echo "$name rc=$? $(tail -n 1 gpurun_out/cfg_$name.log | cut -c1-40,60-175)"; }
run syncbn_b1024 --steps 10 --warmup 4 --batch_size 1024 --sync_bn
run cifar_eager --steps 30 --warmup 10 --batch_size 128 --image_size 32 --num_classes 10 --sync_bn --optimizer adam
run cifar_graph --steps 30 --warmup 10 --batch_size 128 --image_size 32 --num_classes 10 --sync_bn --optimizer adam --graph
run resnet152 --steps 8 --warmup 3 --batch_size 256 --model resnet152
run resnet18 --steps 10 --warmup 4 --batch_size 512 --model resnet18
run vit_b256 --steps 10 --warmup 4 --batch_size 256 --model vit_b_16 --optimizer adamw
run resnet50_b256 --steps 20 --warmup 5 --batch_size 256
