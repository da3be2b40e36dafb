#!/bin/bash
# Short last batch of an epoch under hipGraph replay: the odd-sized batch runs eagerly between replays (BN running
# statistics, Adam's device-side step counter and the loader's ring are shared).  Graph and eager runs of the same
# job must print the same loss curve.   /usr/local/graft/bin/gpurun --timeout 600 -- 'bash bench/short_batch_check.sh'
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/short_batch
rm -rf $OUT && mkdir -p $OUT && cd $OUT
export PYTHONPATH=$R MASTER_ADDR=127.0.0.1
ARGS="--dataset synthetic-cifar --dataset_len 2000 --num_classes 100 --epochs 3 --profile none --seed 0 --model resnet18"
MASTER_PORT=29671 timeout -k 10 200 python -u $R/main.py $ARGS --summary_json g.json --log_dir g > g.log 2>&1 || { tail -5 g.log; exit 1; }
MASTER_PORT=29672 timeout -k 10 200 python -u $R/main.py $ARGS --no_graph --summary_json e.json --log_dir e > e.log 2>&1 || { tail -5 e.log; exit 1; }
python - <<'PY'
import json, re
def losses(p): return [(m.group(1), m.group(2), float(m.group(3))) for m in re.finditer(r"Epoch: (\d+) step: (\d+) loss: ([-+0-9.eE]+)", open(p).read())]
g, e = losses("g.log"), losses("e.log")
sg = json.load(open("g.json")); se = json.load(open("e.json"))
print("graph run captured:", sg.get("hip_graph"), " eager run captured:", se.get("hip_graph"), " steps:", sg.get("steps"), se.get("steps"))
worst = 0.0
for a, b in zip(g, e):
    assert a[:2] == b[:2]
    worst = max(worst, abs(a[2] - b[2]) / max(1.0, abs(b[2])))
    print("epoch %s step %s  graph %.4f  eager %.4f" % (a[0], a[1], a[2], b[2]))
print("worst relative difference: %.4f" % worst)
assert len(g) == len(e) >= 6 and sg.get("hip_graph") and not se.get("hip_graph")
assert g[-1][2] < 0.5 * g[0][2] and worst < 0.25
print("OK")
PY
