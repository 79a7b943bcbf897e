#!/bin/bash
# Round 5, batch Q: the native front after lasr_front_stop / in-call accounting: its GPU tests + the server tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5q; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_server.py -m gpu -q -x > $O/pytest_sel.txt 2>&1; echo "rc=$?" >> $O/pytest_sel.txt; tail -5 $O/pytest_sel.txt
