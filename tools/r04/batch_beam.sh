#!/bin/bash
# Round 4: the beam configs with / without the pump thread, by steps in flight and pump group size
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4g; mkdir -p $O
B="python bench.py --gpus 1 --no-cpu-baseline --no-extras --check-rows 0"
C4="$B --model cfg5 --dtype bf16 --streams 128 --beam 8 --steps 8 --warmup 2"
for d in 3 4 6; do
  LASR_PUMP=0 $C4 --depth $d > $O/cfg5_beam8_pump0_d$d.json 2> $O/cfg5_beam8_pump0_d$d.err
  $C4 --depth $d > $O/cfg5_beam8_pump1_d$d.json 2> $O/cfg5_beam8_pump1_d$d.err
done
for g in 2 4; do LASR_PUMP_G=$g $C4 --depth 3 > $O/cfg5_beam8_pump1_g${g}_d3.json 2> $O/cfg5_beam8_pump1_g${g}_d3.err; done
C2="$B --dtype bf16 --beam 4 --steps 10 --warmup 3"
for g in 2 4 6; do LASR_PUMP_G=$g $C2 > $O/cfg2_beam4_pump1_g$g.json 2> $O/cfg2_beam4_pump1_g$g.err; done
LASR_PUMP=0 $C2 > $O/cfg2_beam4_pump0.json 2> $O/cfg2_beam4_pump0.err
python tools/r04/summ.py $O/*.json
