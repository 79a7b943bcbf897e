#!/bin/bash
# Round 4, A/B: non-temporal loads of the encoder cells' weights (LASR_CELL_NT=1)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4f; mkdir -p $O
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras"
for rep in 1 2; do
  $B > $O/f32_nt0_$rep.json 2> $O/f32_nt0_$rep.err
  LASR_CELL_NT=1 $B > $O/f32_nt1_$rep.json 2> $O/f32_nt1_$rep.err
done
$B --dtype bf16 > $O/bf16_nt0.json 2> $O/bf16_nt0.err
LASR_CELL_NT=1 $B --dtype bf16 > $O/bf16_nt1.json 2> $O/bf16_nt1.err
python tools/r04/summ.py $O/*.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4f/*.json")):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], "isolated cell us", d["roofline"].get("launch_us_isolated"))
    except Exception as e: print(f, e)
PY
