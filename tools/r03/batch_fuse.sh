#!/bin/bash
# fused logits + select (EpiLogitsSel): parity tests, then A/B of the headline
mkdir -p gpurun_out/fuse
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/fuse/pytest.txt
cat gpurun_out/fuse/pytest.txt
for rep in 1 2; do
  for f in 1 0; do
    LASR_FUSE_SELECT=$f timeout 120 python bench.py --steps 40 --no-cpu-baseline --no-extras > gpurun_out/fuse/bench_f${f}_r${rep}.json 2> gpurun_out/fuse/bench_f${f}_r${rep}.err
    python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/fuse/bench_f${f}_r${rep}.json").read().strip().splitlines()[-1])
    print("fuse=${f} rep=${rep}", d["value"], d["roofline"].get("launch_us"), d["stage_ms_per_model_step"]["decode_iters"], d["latency_ms"]["p50_model_chunk"], d.get("tokens_equal"))
except Exception as e:
    print("fuse=${f} rep=${rep} failed", e)
PY
  done
done
