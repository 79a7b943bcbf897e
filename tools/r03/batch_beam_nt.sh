#!/bin/bash
# k_beam_select with 512 threads per stream (V <= 2048) against 1024: parity subset, then A/B
timeout 400 python -m pytest tests/test_gpu_beam.py tests/test_gpu_round2.py -m gpu -q -x 2>&1 | tail -3
run2() { env "$@" timeout 300 python bench.py --dtype bf16 --beam 4 --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('cfg2b4', sys.argv[1:], d['value'], d['stage_ms_per_model_step']['decode_iters'], d['latency_ms']['p50_model_chunk'])" "$@"; }
run4() { env "$@" timeout 300 python bench.py --model cfg5 --dtype bf16 --streams 128 --beam 8 --steps 8 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('cfg5b8', sys.argv[1:], d['value'], d['stage_ms_per_model_step']['decode_iters'], d['latency_ms']['p50_model_chunk'])" "$@"; }
run2 LASR_BEAM_NT=512
run2 LASR_BEAM_NT=1024
run2 LASR_BEAM_NT=512
run2 LASR_BEAM_NT=1024
run4 LASR_BEAM_NT=512
run4 LASR_BEAM_NT=1024
