#!/bin/bash
# Round 4, A/B of the hoisted predictor (recurrent products out of the decode iteration's dependent chain): LASR_HOIST=0 / 1
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4d; mkdir -p $O
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras"
for rep in 1 2; do
  LASR_HOIST=0 $B > $O/f32_hoist0_$rep.json 2> $O/f32_hoist0_$rep.err
  $B > $O/f32_hoist1_$rep.json 2> $O/f32_hoist1_$rep.err
done
LASR_HOIST=0 $B --dtype bf16 > $O/bf16_hoist0.json 2> $O/bf16_hoist0.err
$B --dtype bf16 > $O/bf16_hoist1.json 2> $O/bf16_hoist1.err
LASR_HOIST=0 $B --model cfg5 --dtype bf16 --streams 128 > $O/cfg5_greedy_hoist0.json 2> $O/cfg5_greedy_hoist0.err
$B --model cfg5 --dtype bf16 --streams 128 > $O/cfg5_greedy_hoist1.json 2> $O/cfg5_greedy_hoist1.err
python tools/r04/summ.py $O/*.json
