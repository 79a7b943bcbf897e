#!/bin/bash
# Round 5, batch X: three switches once more on the final build (deferred append on, 20 in flight): f32 main-stream cell graph,
# pump groups of 3, 8 waves per cell workgroup; 100-step runs, interleaved, 3 rounds
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5x; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2 3; do
  for v in "base:X=0" "mg:LASR_MAIN_GRAPH=1" "g3:LASR_PUMP_G=3" "nw8:LASR_CELL_NW=8"; do
    n=${v%%:*}; e=${v#*:}
    env $e timeout 200 python bench.py --gpus 1 --steps 100 --warmup 5 --no-cpu-baseline --no-extras --sustained-s 0 --check-rows 8 2>/dev/null | tail -1 > $O/f32_${n}_$i.json
  done
done
python - <<'PY'
import json, glob, collections
acc = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r5x/*.json")):
    try:
        j = json.loads(open(f).read())
        k = f.split("/")[-1].rsplit("_", 1)[0]
        acc[k].append((j["value"], j["latency_ms"]["p50_model_chunk"], j.get("tokens_equal"), j.get("iterations_per_model_step")))
    except Exception as e:
        print(f, "ERR", e)
for k in sorted(acc):
    v = acc[k]
    print(k, "values", [round(x[0] / 1000, 2) for x in v], "mean", round(sum(x[0] for x in v) / len(v) / 1000, 2), "p50", round(sum(x[1] for x in v) / len(v), 2), "iters", v[0][3], "tok_eq", all(x[2] for x in v))
PY
