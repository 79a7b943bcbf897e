#!/bin/bash
# Round 5, batch T: steps in flight, finer: depth 14 / 16 / 18 / 20 / 22 in the driver's form (20 steps) and at 100 steps, interleaved, 4 rounds
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5t; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2 3 4; do
  for d in 20 16 14 18 22; do
    timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --depth $d --no-cpu-baseline --no-extras --sustained-s 0 --check-rows 8 2>/dev/null | tail -1 > $O/s20_d${d}_$i.json
    timeout 200 python bench.py --gpus 1 --steps 100 --warmup 5 --depth $d --no-cpu-baseline --no-extras --sustained-s 0 --check-rows 8 2>/dev/null | tail -1 > $O/s100_d${d}_$i.json
  done
done
python - <<'PY'
import json, glob, collections
acc = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r5t/*.json")):
    try:
        j = json.loads(open(f).read())
        k = f.split("/")[-1].rsplit("_", 1)[0]
        acc[k].append((j["value"], j["latency_ms"]["p50_model_chunk"], j.get("iterations_per_model_step"), j.get("tokens_equal")))
    except Exception as e:
        print(f, "ERR", e)
for k in sorted(acc):
    v = acc[k]
    print(k, "values", [round(x[0] / 1000, 2) for x in v], "mean", round(sum(x[0] for x in v) / len(v) / 1000, 2), "p50", round(sum(x[1] for x in v) / len(v), 2), "iters", v[0][2], "tok_eq", all(x[3] for x in v))
PY
