#!/bin/bash
# Round 5, batch M: the GPU suite three times in fresh processes (flake check of the round's new tests), then under LASR_POISON=1
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5m; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2 3; do
  timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu_$i.txt 2>&1; echo "rc=$?" >> $O/pytest_gpu_$i.txt; tail -3 $O/pytest_gpu_$i.txt
done
LASR_POISON=1 timeout 1500 python -m pytest tests/test_gpu_round5.py tests/test_gpu_parity.py tests/test_gpu_round3.py -q -x > $O/pytest_poison.txt 2>&1; echo "rc=$?" >> $O/pytest_poison.txt; tail -3 $O/pytest_poison.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
