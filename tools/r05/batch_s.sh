#!/bin/bash
# Round 5, batch S: steps in flight for the DRIVER's form (--steps 20 --warmup 5: fill and drain of the pipeline are inside a 30 ms
# timed region): depth 12 / 16 / 20 / 25, interleaved, 3 rounds
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5s; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2 3; do
  for d in 20 12 16 25; do
    timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --depth $d --no-cpu-baseline --no-extras --sustained-s 0 --check-rows 8 2>/dev/null | tail -1 > $O/f32_d${d}_$i.json
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5s/*.json")):
    try:
        j = json.loads(open(f).read())
        print(f.split("/")[-1], j["value"], j["ms_per_step"], j["latency_ms"]["p50_model_chunk"], j.get("iterations_per_model_step"), j.get("tokens_equal"))
    except Exception as e:
        print(f, "ERR", e)
PY
