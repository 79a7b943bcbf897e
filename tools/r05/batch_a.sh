#!/bin/bash
# Round 5, batch A: (i) the HIP-event pair per model step inside the timed region (bench.py --cell-prof-in-timed 3, the round-4
# default) against none; (ii) the decode throttle LASR_DEC_MIN_ROWS (a group waits for the next encoder when few rows hold frames)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5a; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --gpus 1 --no-cpu-baseline --no-extras --steps 40 --warmup 5"
run() { n=$1; shift; timeout 300 "$@" > $O/$n.json 2> $O/$n.err || echo "rc $? $n" >> $O/failures.txt; }
run ev3_a $B --cell-prof-in-timed 3
run ev0_a $B --cell-prof-in-timed 0
run ev3_b $B --cell-prof-in-timed 3
run ev0_b $B --cell-prof-in-timed 0
for t in 4 8 16 24 32 48; do
  LASR_DEC_MIN_ROWS=$t run thr${t} $B --cell-prof-in-timed 0
done
for g in 1 3; do
  LASR_PUMP_G=$g LASR_DEC_MIN_ROWS=16 run thr16_g$g $B --cell-prof-in-timed 0
  LASR_PUMP_G=$g run thr0_g$g $B --cell-prof-in-timed 0
done
LASR_DEC_MIN_ROWS=16 run thr16_bf16 $B --cell-prof-in-timed 0 --dtype bf16
run thr0_bf16 $B --cell-prof-in-timed 0 --dtype bf16
LASR_DEC_MIN_ROWS=16 timeout 300 python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py -q -x > $O/pytest_thr16.txt 2>&1
python tools/r05/summ.py $O/*.json | tee $O/summary.txt
tail -3 $O/pytest_thr16.txt
