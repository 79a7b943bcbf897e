#!/bin/bash
# joint encoder-half GEMM on a side stream (LASR_PE_SIDE=1): parity, then A/B
mkdir -p gpurun_out/peside
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], d["value"], d["roofline"].get("launch_us"), d["stage_ms_per_model_step"]["decode_iters"], d["latency_ms"]["p50_model_chunk"], d.get("tokens_equal"))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
LASR_PE_SIDE=1 timeout 300 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round2.py -m gpu -x -q 2>&1 | tail -5
for rep in 1 2 3; do
  for f in 1 0; do
    LASR_PE_SIDE=$f timeout 120 python bench.py --steps 40 --no-cpu-baseline --no-extras > gpurun_out/peside/f32_p${f}_r${rep}.json 2> gpurun_out/peside/f32_p${f}_r${rep}.err
    show "f32 pe_side=$f rep=$rep" gpurun_out/peside/f32_p${f}_r${rep}.json
  done
done
for f in 1 0; do
  LASR_PE_SIDE=$f timeout 120 python bench.py --steps 40 --dtype bf16 --no-cpu-baseline --no-extras > gpurun_out/peside/bf16_p${f}.json 2> gpurun_out/peside/bf16_p${f}.err
  show "bf16 pe_side=$f" gpurun_out/peside/bf16_p${f}.json
  LASR_PE_SIDE=$f timeout 120 python bench.py --steps 16 --no-cpu-baseline --no-extras --trace gpurun_out/peside/trace_p${f}.txt > gpurun_out/peside/tr_p${f}.json 2> gpurun_out/peside/tr_p${f}.err
  python tools/stream_timeline.py gpurun_out/peside/trace_p${f}.txt 2>&1 | head -2
done
