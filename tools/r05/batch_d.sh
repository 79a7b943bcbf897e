#!/bin/bash
# Round 5, batch D: full GPU suite on the cleaned-up build (26 switches removed, debug header split), then the round's profile set
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5d; mkdir -p $O
export TMPDIR=/tmp
rm -f gpurun_out/parity_counts.json gpurun_out/served_rate.json
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "rc=$?" >> $O/pytest_gpu.txt; tail -6 $O/pytest_gpu.txt
cp gpurun_out/parity_counts.json gpurun_out/served_rate.json $O/ 2>/dev/null
bash tools/profile_round5.sh
