#!/bin/bash
# Round 5, batch U: pump nap (LASR_PUMP_NAP_PCT 0 / 40 / 60 / 75): rate, p50, host cores busy per rank; 100-step runs, interleaved, 3 rounds
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5u; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_round5.py -m gpu -q -x -k "PUMP_NAP or deferred" > $O/pytest_sel.txt 2>&1; echo "rc=$?" >> $O/pytest_sel.txt; tail -2 $O/pytest_sel.txt
for i in 1 2 3; do
  for p in 0 60 40 75; do
    LASR_PUMP_NAP_PCT=$p timeout 200 python bench.py --gpus 1 --steps 100 --warmup 5 --no-cpu-baseline --no-extras --sustained-s 0 --check-rows 8 2>/dev/null | tail -1 > $O/f32_p${p}_$i.json
    LASR_PUMP_NAP_PCT=$p timeout 200 python bench.py --gpus 1 --dtype bf16 --steps 100 --warmup 5 --no-cpu-baseline --no-extras --sustained-s 0 --check-rows 8 2>/dev/null | tail -1 > $O/bf16_p${p}_$i.json
  done
done
python - <<'PY'
import json, glob, collections
acc = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r5u/*.json")):
    try:
        j = json.loads(open(f).read())
        k = f.split("/")[-1].rsplit("_", 1)[0]
        acc[k].append((j["value"], j["latency_ms"]["p50_model_chunk"], j["per_rank"][0].get("host_cores_busy"), j.get("tokens_equal"), j.get("iterations_per_model_step")))
    except Exception as e:
        print(f, "ERR", e)
for k in sorted(acc):
    v = acc[k]
    print(k, "values", [round(x[0] / 1000, 2) for x in v], "mean", round(sum(x[0] for x in v) / len(v) / 1000, 2), "p50", round(sum(x[1] for x in v) / len(v), 2), "cores", [x[2] for x in v], "iters", v[0][4], "tok_eq", all(x[3] for x in v))
PY
