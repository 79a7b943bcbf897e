#!/bin/bash
# same-box A/B of one switch with every chunk in pageable host memory: usage ab_env_hostpcm.sh VAR=value <out_dir> <rounds>
SW=$1; OUT=$2; N=${3:-3}
mkdir -p $OUT
for i in $(seq 1 $N); do
  timeout 300 python bench.py --host-pcm --no-cpu-baseline --other-configs 0 --no-extras 2>/dev/null | tail -1 > $OUT/host_off_$i.json
  env $SW timeout 300 python bench.py --host-pcm --no-cpu-baseline --other-configs 0 --no-extras 2>/dev/null | tail -1 > $OUT/host_on_$i.json
done
python - <<PY
import json,glob
for tag in ("off","on"):
    v=[]
    for f in sorted(glob.glob("$OUT/host_%s_*.json"%tag)):
        try:
            j=json.loads(open(f).read()); v.append((round(j["value"]), round(j["sustained"]["value"]), j["tokens_equal"], j["per_rank"][0]["host_us_per_model_step"]["push"]))
        except Exception as e: v.append(("?",str(e)[:40]))
    print(tag,v)
PY
