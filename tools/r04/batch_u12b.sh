#!/bin/bash
# tiling D: with / without the layer wavefront, 4 / 8 waves
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4k; mkdir -p $O
B="python bench.py --gpus 1 --no-cpu-baseline --no-extras --model cfg5 --dtype bf16 --streams 128"
for w in 0 1; do for nw in 4 8; do
  LASR_ENC_WAVE=$w LASR_CELL_NW=$nw $B --steps 20 --warmup 5 --depth 6 > $O/cfg5_greedy_D_wave${w}_nw$nw.json 2> $O/cfg5_greedy_D_wave${w}_nw$nw.err
  LASR_ENC_WAVE=$w LASR_CELL_NW=$nw $B --check-rows 0 --beam 8 --steps 8 --warmup 2 --depth 6 > $O/cfg5_beam8_D_wave${w}_nw$nw.json 2> $O/cfg5_beam8_D_wave${w}_nw$nw.err
done; done
LASR_ENC_U12=0 LASR_ENC_WAVE=0 $B --steps 20 --warmup 5 --depth 6 > $O/cfg5_greedy_C_wave0.json 2> $O/cfg5_greedy_C_wave0.err
python tools/r04/summ.py $O/*.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4k/*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]; print(f.split("/")[-1], "frac", r["frac"], "launch_us", r["launch_us"], "cells/launch", r["cells_per_launch"], "isolated", r.get("launch_us_isolated"))
    except Exception as e: print(f, e)
PY
