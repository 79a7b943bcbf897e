# round 6: full GPU suite (writes the complete gpurun_out/parity_counts.json) + the driver-form line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6z; mkdir -p $O; cd $R
rm -f gpurun_out/parity_counts.json
( time timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -12 ) > $O/pytest_full.txt 2>&1
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driverform.json 2> $O/bench_driverform.err ) 2> $O/bench_driverform.time
cat $O/pytest_full.txt $O/bench_driverform.time; cut -c1-400 $O/bench_driverform.json
