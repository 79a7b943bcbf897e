#!/bin/bash
# same-box A/B of two builds with every chunk handed over in pageable host memory (bench.py --host-pcm): usage ab_hostpcm.sh <old.so> <new.so> <out_dir> <rounds>
OLD=$1; NEW=$2; OUT=$3; N=${4:-3}
mkdir -p $OUT
for i in $(seq 1 $N); do for tag in old new; do lib=$OLD; [ $tag = new ] && lib=$NEW
  LASR_LIB=$lib timeout 300 python bench.py --host-pcm --no-cpu-baseline --other-configs 0 --no-extras 2>/dev/null | tail -1 > $OUT/host_${tag}_$i.json; done; done
python - <<PY
import json,glob
for tag in ("old","new"):
    v=[]
    for f in sorted(glob.glob("$OUT/host_%s_*.json"%tag)):
        try:
            j=json.loads(open(f).read()); v.append((round(j["value"]), round(j["sustained"]["value"]), j["tokens_equal"], j["per_rank"][0]["host_us_per_model_step"]["push"]))
        except Exception as e: v.append(("?",str(e)[:40]))
    print(tag,v)
PY
