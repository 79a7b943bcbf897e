R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4d6; mkdir -p $O; cd $R
B="timeout 300 python3 bench.py --no-cpu-baseline --no-extras --check-rows 0 --model cfg5 --dtype bf16 --streams 128 --beam 8"
for i in 1 2 3; do $B --steps 8 --warmup 2 --depth 6 > $O/d6_$i.json 2>/dev/null; done
$B --steps 24 --warmup 4 --depth 6 > $O/d6_long.json 2>/dev/null
$B --steps 8 --warmup 2 > $O/d3.json 2>/dev/null
LASR_BEAM_CARRY=1 $B --steps 8 --warmup 2 --depth 6 > $O/d6_carry1.json 2>/dev/null
for f in $O/*.json; do python3 -c "
import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], d['value'], d['latency_ms']['p50_model_chunk'], d['latency_ms']['p95_model_chunk'])"; done
