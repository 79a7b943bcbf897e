#!/bin/bash
# k_ln_tile's store phase on 1 / 2 / 4 / 8 z-slices (bit-identical outputs)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4o; mkdir -p $O
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras"
for rep in 1 2; do for z in 1 2 4 8; do LASR_LN_Z=$z $B > $O/f32_lnz${z}_$rep.json 2> $O/e.err; done; done
for z in 1 4; do LASR_LN_Z=$z $B --dtype bf16 > $O/bf16_lnz$z.json 2>> $O/e.err; done
python tools/r04/summ.py $O/*.json
