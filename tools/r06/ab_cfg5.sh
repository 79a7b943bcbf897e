#!/bin/bash
# same-box A/B of two builds on the cfg5 shapes (beam 8 and greedy, 128 streams): usage ab_cfg5.sh <old.so> <new.so> <out_dir> <rounds>
OLD=$1; NEW=$2; OUT=$3; N=${4:-3}
mkdir -p $OUT
C="--no-cpu-baseline --other-configs 0 --no-extras --sustained-s 0"
for i in $(seq 1 $N); do for tag in old new; do lib=$OLD; [ $tag = new ] && lib=$NEW
  LASR_LIB=$lib timeout 300 python bench.py $C --model cfg5 --dtype bf16 --beam 8 --streams 128 --steps 4 --warmup 1 2>/dev/null | tail -1 > $OUT/cfg5b8_${tag}_$i.json
  LASR_LIB=$lib timeout 300 python bench.py $C --model cfg5 --dtype bf16 --streams 128 --steps 8 --warmup 2 2>/dev/null | tail -1 > $OUT/cfg5g_${tag}_$i.json
done; done
python - <<PY
import json,glob
for cfg in ("cfg5b8","cfg5g"):
  for tag in ("old","new"):
    v=[]
    for f in sorted(glob.glob("$OUT/%s_%s_*.json"%(cfg,tag))):
        try:
            j=json.loads(open(f).read()); v.append((round(j["value"]), j["latency_ms"]["p50_model_chunk"], j["roofline"]["launch_us"], j.get("tokens_equal")))
        except Exception as e: v.append(("?",str(e)[:40]))
    print(cfg,tag,v)
PY
