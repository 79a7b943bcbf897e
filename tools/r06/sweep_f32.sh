#!/bin/bash
# configs[1] knob sweep on one box (bench.py --no-extras, two runs per setting): usage sweep_f32.sh <out_dir>
OUT=$1; mkdir -p $OUT
COMMON="--no-cpu-baseline --other-configs 0 --no-extras"
run() { tag=$1; shift; for i in 1 2; do env "$@" timeout 300 python bench.py $COMMON 2>/dev/null | tail -1 > $OUT/${tag}_$i.json; done; }
run base LASR_VERBOSE=0
run la1 LASR_LOOKAHEAD=1
run la3 LASR_LOOKAHEAD=3
run la4 LASR_LOOKAHEAD=4
run g4 LASR_PUMP_G=4
run g6 LASR_PUMP_G=6
run nw8 LASR_CELL_NW=8
run mg1 LASR_MAIN_GRAPH=1
run base2 LASR_VERBOSE=0
python - <<PY
import json,glob
import collections
r=collections.OrderedDict()
for f in sorted(glob.glob("$OUT/*.json")):
    tag=f.split("/")[-1].rsplit("_",1)[0]
    try:
        j=json.loads(open(f).read()); r.setdefault(tag,[]).append((round(j["value"]), round(j["sustained"]["value"]), j["sustained"]["p95_model_chunk_ms"], j["stream_timeline_us_per_model_step"].get("decode_iterations_per_step")))
    except Exception as e: r.setdefault(tag,[]).append(str(e)[:30])
for k,v in r.items(): print(k,v)
PY
