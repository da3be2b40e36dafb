# same-box A/B of the non-temporal size threshold (PDT_STREAM_NT_MIN_MB): bash bench/micro/nt_ab.sh "32 128 0" [bench args]
vals=$1; shift
for v in $vals $vals; do PDT_STREAM_NT_MIN_MB=$v timeout -k 10 200 python bench.py --steps 10 --warmup 4 "$@" 2>/dev/null | cut -c60-175 | sed "s/^/nt_min_mb=$v /" || exit 1; done
