#!/bin/bash
# Round 5, batch C: (i) timing experiment -- token ring / step marks in device memory instead of pinned host memory
# (LASR_EXP_TOKDEV=1: tokens never reach the host; what the zero-copy stores over PCIe cost the selection kernels);
# (ii) the full GPU suite on the round's build (default switches)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c; mkdir -p $O
export TMPDIR=/tmp
run() { n=$1; shift; timeout 300 "$@" > $O/$n.json 2> $O/$n.err || echo "rc $? $n" >> $O/failures.txt; }
B="python bench.py --gpus 1 --no-cpu-baseline --no-extras --steps 40 --warmup 5 --sustained-s 0"
for i in a b; do
  run tokhost_$i $B
  LASR_EXP_TOKDEV=1 run tokdev_$i $B --check-rows 0
done
LASR_EXP_TOKDEV=1 LASR_PUMP_G=3 run tokdev_g3 $B --check-rows 0
LASR_PUMP_G=3 run tokhost_g3 $B
run bf16_tokhost $B --dtype bf16
LASR_EXP_TOKDEV=1 run bf16_tokdev $B --dtype bf16 --check-rows 0
python tools/r05/summ.py $O/*.json | tee $O/summary.txt
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
cat $O/failures.txt 2>/dev/null
