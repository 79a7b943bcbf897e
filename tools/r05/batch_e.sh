#!/bin/bash
# Round 5, batch E: the native serving front -- its GPU tests, then the served rate against the Python scheduler's
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5e; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round5.py -q -x -k "native_front" -s > $O/pytest_front.txt 2>&1; echo "rc=$?" >> $O/pytest_front.txt; tail -25 $O/pytest_front.txt
cp gpurun_out/served_rate_native.json $O/ 2>/dev/null
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_check.json 2> $O/bench_check.err; python tools/r05/summ.py $O/bench_check.json
