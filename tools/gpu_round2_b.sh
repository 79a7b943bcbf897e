set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2b; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo rc=$? >> $O/bench_driver.err
timeout 300 python3 bench.py --no-cpu-baseline --no-extras --dtype bf16 > $O/bench_bf16.json 2> $O/bench_bf16.err
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_driver -o kt -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/kt_driver.json 2>$O/kt_driver.err
ls -R $O | head -40
cat $O/bench_driver.json | head -c 1500; tail -3 $O/bench_driver.err
