#!/bin/bash
# Round 5, batch W (last): the GPU suite on the final build, then the driver's command once more (it now reads the traced cell
# average of the final profile batch from profiles/cell_rocprof.json)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5w; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "rc=$?" >> $O/pytest_gpu.txt; tail -3 $O/pytest_gpu.txt
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driverform.json 2> $O/bench_driverform.err; echo rc=$? >> $O/bench_driverform.err
python - <<'PY'
import json
j = json.load(open("gpurun_out/r5w/bench_driverform.json"))
print(j["value"], j["ms_per_step"], j["latency_ms"]["p50_model_chunk"], j["tokens_equal"], j["roofline"]["frac"], j["roofline"]["launch_us"], j["roofline"]["launch_us_rocprof"], j["sustained"]["value"], j["cpu_baseline"]["value"])
PY
