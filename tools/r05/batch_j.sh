#!/bin/bash
# Round 5, batch J: with 20 steps in flight -- lookahead 3 (a backlogged row consumes 3 blank frames per iteration) and pump groups of 3
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5j; mkdir -p $O
export TMPDIR=/tmp
run() { n=$1; shift; timeout 300 "$@" > $O/$n.json 2> $O/$n.err || echo "rc $? $n" >> $O/failures.txt; }
B="python bench.py --gpus 1 --no-cpu-baseline --no-extras --steps 40 --warmup 5 --sustained-s 0 --depth 20"
for i in a b; do
  run base_$i $B
  LASR_LOOKAHEAD=3 run la3_$i $B
  LASR_PUMP_G=3 run g3_$i $B
  LASR_PUMP_G=1 run g1_$i $B
done
LASR_LOOKAHEAD=3 run la3_bf16 $B --dtype bf16
run base_bf16 $B --dtype bf16
LASR_PUMP_G=3 run g3_bf16 $B --dtype bf16
python - <<'PY'
import json, glob
for p in sorted(glob.glob("gpurun_out/r5j/*.json")):
    try: d = json.loads(open(p).read().strip().splitlines()[-1])
    except Exception: continue
    print(p.split("/")[-1], d["value"], "p50", d["latency_ms"]["p50_model_chunk"], "iters", d["iterations_per_model_step"], d.get("tokens_equal"))
PY
