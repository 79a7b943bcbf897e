R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3p; mkdir -p $O
cd $R
timeout 900 python -u -m pytest tests -m gpu -x -q 2>&1 | tail -25
B="timeout 200 python3 bench.py --no-cpu-baseline --no-extras --check-rows 0"
for a in "--dtype bf16 --beam 4 --steps 10 --warmup 3" "--model cfg5 --dtype bf16 --streams 128 --beam 8 --steps 8 --warmup 2"; do $B $a 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['latency_ms']['p50_model_chunk'], d['stage_ms_per_model_step']['decode_iters'])"; done
