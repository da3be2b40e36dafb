# A/B of an environment switch on ONE box (box-to-box variance is +-1 %): bash bench/micro/env_ab.sh VAR "1 0" [bench args]
var=$1; vals=$2; shift 2
for v in $vals $vals; do env $var=$v timeout -k 10 200 python bench.py --steps 12 --warmup 4 "$@" 2>/dev/null | cut -c60-130 | sed "s/^/$var=$v /" || exit 1; done
