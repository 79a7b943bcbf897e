#!/bin/bash
# Round 4, A/B of the native pump thread (decode groups launched by a library thread instead of the API calls).
# Driver-form bench (f32 configs[1]) with LASR_PUMP=0 / 1 and the pump's group size; bf16 and beam legs.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c; mkdir -p $O
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras"
for rep in 1 2; do
  LASR_PUMP=0 $B > $O/f32_pump0_$rep.json 2> $O/f32_pump0_$rep.err
  for g in 2 3 4; do LASR_PUMP_G=$g $B > $O/f32_pump1_g${g}_$rep.json 2> $O/f32_pump1_g${g}_$rep.err; done
done
LASR_PUMP=0 $B --dtype bf16 > $O/bf16_pump0.json 2> $O/bf16_pump0.err
for g in 2 3 4; do LASR_PUMP_G=$g $B --dtype bf16 > $O/bf16_pump1_g$g.json 2> $O/bf16_pump1_g$g.err; done
LASR_PUMP=0 $B --dtype bf16 --beam 4 > $O/beam4_pump0.json 2> $O/beam4_pump0.err
$B --dtype bf16 --beam 4 > $O/beam4_pump1.json 2> $O/beam4_pump1.err
python tools/r04/summ.py $O/*.json
