set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2a; mkdir -p $O
cd $R
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo rc=$? >> $O/bench_driver.err
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver2.json 2>> $O/bench_driver.err
timeout 400 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo rc=$? >> $O/bench_default.err
LASR_BENCH_BACKEND=gloo LASR_BENCH_SAME_GPU=1 timeout 300 python3 bench.py --gpus 2 --steps 40 --warmup 10 --no-cpu-baseline > $O/bench_2rank.json 2> $O/bench_2rank.err; echo rc=$? >> $O/bench_2rank.err
timeout 100 python3 bench.py --gpus 2 --steps 4 --warmup 2 > $O/bench_refuse.json 2> $O/bench_refuse.err; echo rc=$? >> $O/bench_refuse.err
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_driver -o kt -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/kt_driver.json 2>$O/kt_driver.err
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_default -o kt -- python3 $R/bench.py --no-cpu-baseline --no-extras > $O/kt_default.json 2>$O/kt_default.err
ls -R $O | head -60
cat $O/bench_driver.json; tail -3 $O/bench_driver.err
