#!/bin/bash
# same-box A/B of two builds on configs[1] only: usage ab_f32.sh <old.so> <new.so> <out_dir> <rounds>
OLD=$1; NEW=$2; OUT=$3; N=${4:-3}
mkdir -p $OUT
for i in $(seq 1 $N); do for tag in old new; do lib=$OLD; [ $tag = new ] && lib=$NEW
  LASR_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --other-configs 0 --no-extras 2>/dev/null | tail -1 > $OUT/f32_${tag}_$i.json; done; done
python - <<PY
import json,glob
for tag in ("old","new"):
    v=[]
    for f in sorted(glob.glob("$OUT/f32_%s_*.json"%tag)):
        try:
            j=json.loads(open(f).read()); v.append((round(j["value"]), round(j["sustained"]["value"]), j["tokens_equal"], j["roofline"]["launch_us"]))
        except Exception as e: v.append(("?",str(e)[:40]))
    print(tag,v)
PY
