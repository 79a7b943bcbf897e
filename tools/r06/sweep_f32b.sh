#!/bin/bash
OUT=$1; mkdir -p $OUT
COMMON="--no-cpu-baseline --other-configs 0 --no-extras"
run() { tag=$1; shift; for i in 1 2; do env "$@" timeout 300 python bench.py $COMMON $EXTRA 2>/dev/null | tail -1 > $OUT/${tag}_$i.json; done; }
EXTRA="" run base LASR_VERBOSE=0
EXTRA="" run wave1 LASR_ENC_WAVE=1
EXTRA="--depth 20" run d20 LASR_VERBOSE=0
EXTRA="--depth 22" run d22 LASR_VERBOSE=0
EXTRA="--depth 16" run d16 LASR_VERBOSE=0
EXTRA="" run base2 LASR_VERBOSE=0
python - <<PY
import json,glob,collections
r=collections.OrderedDict()
for f in sorted(glob.glob("$OUT/*.json")):
    tag=f.split("/")[-1].rsplit("_",1)[0]
    try:
        j=json.loads(open(f).read()); r.setdefault(tag,[]).append((round(j["value"]), round(j["sustained"]["value"]), j["sustained"]["p95_model_chunk_ms"], j["latency_ms"]["p50_model_chunk"]))
    except Exception as e: r.setdefault(tag,[]).append(str(e)[:30])
for k,v in r.items(): print(k,v)
PY
