#!/bin/bash
# Round 4: encoder cell tiling D (12 units x 64 rows) on the configs[4] shape: LASR_ENC_U12=0 / 1
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4j; mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round2.py -q -m gpu -x 2>&1 | tail -12) > $O/pytest_round4.txt; tail -5 $O/pytest_round4.txt
B="python bench.py --gpus 1 --no-cpu-baseline --no-extras --model cfg5 --dtype bf16 --streams 128"
for u in 0 1; do
  LASR_ENC_U12=$u $B --steps 20 --warmup 5 --depth 6 > $O/cfg5_greedy_u12_$u.json 2> $O/cfg5_greedy_u12_$u.err
  LASR_ENC_U12=$u $B --check-rows 0 --beam 8 --steps 8 --warmup 2 --depth 6 > $O/cfg5_beam8_d6_u12_$u.json 2> $O/cfg5_beam8_d6_u12_$u.err
  LASR_ENC_U12=$u $B --check-rows 0 --beam 8 --steps 8 --warmup 2 --depth 3 > $O/cfg5_beam8_d3_u12_$u.json 2> $O/cfg5_beam8_d3_u12_$u.err
done
python tools/r04/summ.py $O/*.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4j/*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]; print(f.split("/")[-1], "frac", r["frac"], "launch_us", r["launch_us"], "cells/launch", r["cells_per_launch"], "isolated", r.get("launch_us_isolated"), r["kernel"][:60])
    except Exception as e: print(f, e)
PY
