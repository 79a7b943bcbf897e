#!/bin/bash
# Round 5, batch V: the 8-rank host dry run (8 ranks on ONE GPU, gloo) with and without the pump nap: host cores busy per rank
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5v; mkdir -p $O
export TMPDIR=/tmp
for p in 0 75; do
  LASR_PUMP_NAP_PCT=$p LASR_BENCH_BACKEND=gloo LASR_BENCH_SAME_GPU=1 timeout 300 python3 bench.py --gpus 8 --no-cpu-baseline --no-extras --sustained-s 0 --check-rows 8 2>$O/dry8_p$p.err | tail -1 > $O/dry8_p$p.json
done
python - <<'PY'
import json
for p in (0, 75):
    j = json.load(open(f"gpurun_out/r5v/dry8_p{p}.json"))
    print(p, j["value"], j["tokens_equal"], [r.get("host_cores_busy") for r in j["per_rank"]])
PY
