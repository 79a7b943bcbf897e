#!/bin/bash
# Round 5, batch R: flake check of the final build: the GPU suite twice in fresh processes, smoke
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5r; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2; do
  timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu_$i.txt 2>&1; echo "rc=$?" >> $O/pytest_gpu_$i.txt; tail -3 $O/pytest_gpu_$i.txt
done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
