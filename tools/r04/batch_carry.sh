#!/bin/bash
# Round 4: beam rounds with the non-extended slots carried by k_beam_carry (+ hoisted compaction loads, idle m-groups return at once)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s; mkdir -p $O
(timeout 800 python -m pytest tests/test_gpu_beam.py tests/test_gpu_round2.py tests/test_gpu_lm.py -q -m gpu -x 2>&1 | tail -8) > $O/pytest_beam.txt; tail -3 $O/pytest_beam.txt
B="python bench.py --gpus 1 --no-cpu-baseline --no-extras --check-rows 0"
C4="$B --model cfg5 --dtype bf16 --streams 128 --beam 8 --steps 8 --warmup 2"
C2="$B --dtype bf16 --beam 4 --steps 10 --warmup 3"
for v in 0 1; do
  LASR_BEAM_CARRY=$v $C4 --depth 6 > $O/cfg5_beam8_d6_carry$v.json 2>$O/e.err
  LASR_BEAM_CARRY=$v $C4 --depth 3 > $O/cfg5_beam8_d3_carry$v.json 2>>$O/e.err
  LASR_BEAM_CARRY=$v $C2 > $O/cfg2_beam4_carry$v.json 2>>$O/e.err
done
LASR_BEAM_CARRY=1 $C4 --depth 4 > $O/cfg5_beam8_d4_carry1.json 2>>$O/e.err
python tools/r04/summ.py $O/*.json
