#!/bin/bash
# Round 5, batch I: up to 25 steps in flight (NFLY 16 -> 32): the headline by depth; the depth-limit and parity tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5i; mkdir -p $O
export TMPDIR=/tmp
run() { n=$1; shift; timeout 300 "$@" > $O/$n.json 2> $O/$n.err || echo "rc $? $n" >> $O/failures.txt; }
timeout 600 python -m pytest tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_parity.py -q -x 2>&1 | tail -3
B="python bench.py --gpus 1 --no-cpu-baseline --no-extras --steps 40 --warmup 5 --sustained-s 0"
for i in a b; do for d in 15 18 20 22 25; do run f32_depth${d}_$i $B --depth $d; done; done
for d in 15 20 25; do run bf16_depth$d $B --dtype bf16 --depth $d; done
python - <<'PY'
import json, glob
for p in sorted(glob.glob("gpurun_out/r5i/*.json")):
    try: d = json.loads(open(p).read().strip().splitlines()[-1])
    except Exception: continue
    print(p.split("/")[-1], d["value"], "p50", d["latency_ms"]["p50_model_chunk"], "p95", d["latency_ms"]["p95_model_chunk"], "iters", d["iterations_per_model_step"], d.get("tokens_equal"))
PY
cat $O/failures.txt 2>/dev/null
