#!/bin/bash
# configs[2] (bf16, beam 4): iterations per decode group (LASR_KICK from push/submit, LASR_GROUP while waiting)
run() { env "$@" timeout 300 python bench.py --dtype bf16 --beam 4 --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(sys.argv[1:], d['value'], d['stage_ms_per_model_step']['decode_iters'], d['latency_ms']['p50_model_chunk'], d.get('tokens_equal'))" "$@"; }
run A=base
run LASR_GROUP=2
run LASR_GROUP=4
run LASR_KICK=2
run LASR_KICK=4 LASR_GROUP=4
run LASR_KICK=2 LASR_GROUP=2
run A=base2
