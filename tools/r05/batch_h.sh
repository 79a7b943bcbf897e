#!/bin/bash
# Round 5, batch H: steps in flight of the headline job, three runs each (batch G's single runs: 12 -> 53.7 k, 14 -> 54.8 k, 15 -> 55.5 k)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5h; mkdir -p $O
export TMPDIR=/tmp
run() { n=$1; shift; timeout 300 "$@" > $O/$n.json 2> $O/$n.err || echo "rc $? $n" >> $O/failures.txt; }
B="python bench.py --gpus 1 --no-cpu-baseline --no-extras --steps 40 --warmup 5 --sustained-s 0"
for i in a b c; do for d in 12 13 14 15; do run f32_depth${d}_$i $B --depth $d; done; done
for d in 12 15; do run bf16_depth$d $B --dtype bf16 --depth $d; done
python - <<'PY'
import json, glob
for p in sorted(glob.glob("gpurun_out/r5h/*.json")):
    try: d = json.loads(open(p).read().strip().splitlines()[-1])
    except Exception: continue
    print(p.split("/")[-1], d["value"], "p50", d["latency_ms"]["p50_model_chunk"], "p95", d["latency_ms"]["p95_model_chunk"], "iters", d["iterations_per_model_step"], d.get("tokens_equal"))
PY
