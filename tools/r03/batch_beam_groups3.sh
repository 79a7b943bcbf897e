#!/bin/bash
run2() { env "$@" timeout 300 python bench.py --dtype bf16 --beam 4 --steps 10 --warmup 3 --depth 12 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('cfg2b4 d12', sys.argv[1:], d['value'], d['stage_ms_per_model_step']['decode_iters'], d['latency_ms']['p50_model_chunk'])" "$@"; }
run4() { env "$@" timeout 300 python bench.py --model cfg5 --dtype bf16 --streams 128 --beam 8 --steps 8 --warmup 2 --depth 6 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('cfg5b8 d6', sys.argv[1:], d['value'], d['stage_ms_per_model_step']['decode_iters'], d['latency_ms']['p50_model_chunk'])" "$@"; }
run2 A=base
run2 LASR_GROUP=4
run4 A=base
run4 LASR_GROUP=2
run4 LASR_GROUP=4
