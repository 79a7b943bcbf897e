# round-4 profile set (run under gpurun): GPU tests (monolithic, as the driver runs them, and one fresh process per file), driver-form
# bench + rocprofv3 kernel stats of the same command, the other BASELINE configs, stream timeline, dist legs, soak.
# Summaries are copied to profiles/r04/ by hand.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4final; mkdir -p $O
cd $R
rm -f gpurun_out/parity_counts.json gpurun_out/served_rate.json
timeout 900 python -m pytest tests/ -x -q -m gpu > $O/pytest_monolithic.log 2>&1; echo "rc=$?" >> $O/pytest_monolithic.log; tail -3 $O/pytest_monolithic.log
: > $O/pytest_per_file.txt
for f in tests/test_gpu_*.py; do
  timeout 600 python -m pytest $f -q -m gpu > $O/pf.log 2>&1; echo "$f rc=$? $(tail -1 $O/pf.log)" >> $O/pytest_per_file.txt
done
cat $O/pytest_per_file.txt
cp gpurun_out/parity_counts.json gpurun_out/served_rate.json $O/ 2>/dev/null
timeout 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driverform.json 2> $O/bench_driverform.err; echo rc=$? >> $O/bench_driverform.err
B="timeout 300 python3 bench.py --no-cpu-baseline --no-extras --check-rows 0"
$B --steps 100 --warmup 5 > $O/bench_f32_steps100.json 2>/dev/null
$B --dtype bf16 > $O/bench_bf16.json 2>/dev/null
$B --dtype bf16 --beam 4 --steps 10 --warmup 3 > $O/bench_bf16_beam4.json 2>/dev/null
$B --dtype bf16 --beam 4 --steps 10 --warmup 3 --no-pipeline > $O/bench_bf16_beam4_sync.json 2>/dev/null
$B --model cfg5 --dtype bf16 --streams 128 --depth 6 > $O/bench_cfg5_bf16.json 2>/dev/null
$B --model cfg5 --dtype bf16 --streams 128 --beam 8 --steps 8 --warmup 2 > $O/bench_cfg5_bf16_beam8.json 2>/dev/null
$B --model cfg5 --dtype bf16 --streams 128 --beam 8 --steps 8 --warmup 2 --depth 6 > $O/bench_cfg5_bf16_beam8_depth6.json 2>/dev/null
$B --model cfg5 --dtype bf16 --streams 128 --beam 8 --steps 8 --warmup 2 --no-pipeline > $O/bench_cfg5_bf16_beam8_sync.json 2>/dev/null
$B --no-pipeline > $O/bench_f32_sync.json 2>/dev/null
$B --depth 6 > $O/bench_f32_depth6.json 2>/dev/null
LASR_PUMP=0 $B > $O/bench_f32_pump0.json 2>/dev/null
LASR_BENCH_BACKEND=gloo LASR_BENCH_SAME_GPU=1 timeout 300 python3 bench.py --gpus 2 --no-cpu-baseline --no-extras > $O/bench_2rank_dryrun.json 2> $O/bench_2rank_dryrun.err; echo rc=$? >> $O/bench_2rank_dryrun.err
LASR_BENCH_FORCE_DIST=1 $B --steps 40 > $O/bench_rccl_world1.json 2>/dev/null
$B --trace $O/trace_f32.json > /dev/null 2>&1
python3 tools/stream_timeline.py $O/trace_f32.json > $O/timeline_f32.txt
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_driver -o kt -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --check-rows 0 > $O/kt_driverform.json 2>$O/kt_driver.err
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_beam -o kt -- python3 $R/bench.py --no-cpu-baseline --no-extras --check-rows 0 --dtype bf16 --beam 4 --steps 6 --warmup 2 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_cfg5 -o kt -- python3 $R/bench.py --no-cpu-baseline --no-extras --check-rows 0 --model cfg5 --dtype bf16 --streams 128 --beam 8 --steps 8 --warmup 2 --depth 6 > /dev/null 2>&1
cd $R
python3 tools/rocpd_stats.py $O/kt_driver/kt_results.db $O/kernel_stats_driverform.txt > /dev/null 2>&1
python3 tools/rocpd_gaps.py $O/kt_driver/kt_results.db > $O/kernel_gaps_driverform.txt 2>&1
python3 tools/rocpd_stats.py $O/kt_beam/kt_results.db $O/kernel_stats_bf16_beam4.txt > /dev/null 2>&1
python3 tools/rocpd_stats.py $O/kt_cfg5/kt_results.db $O/kernel_stats_cfg5_bf16_beam8.txt > /dev/null 2>&1
rm -rf $O/kt_driver $O/kt_beam $O/kt_cfg5 $O/trace_f32.json $O/pf.log
# configs[1] + the reference's LM (fp32 / int8-served), serving through the scheduler (trunk form, reset rule)
for lm in fp32 int8; do $B --lm $lm --steps 300 --warmup 50 > $O/bench_lm_$lm.json 2>/dev/null; done
for a in "12 0" "12 1"; do PROF=0 timeout 100 python3 tools/r04/served_profile.py $a 2>/dev/null | grep rep >> $O/served_replay.txt; done
RULE=0 PROF=0 timeout 100 python3 tools/r04/served_profile.py 12 0 2>/dev/null | grep rep >> $O/served_replay.txt
(timeout 300 python tests/soak.py --preamble full --scenario both --iters 1000 --no-dump --out $O/soak_summary.jsonl 2>&1 | tail -2) > $O/soak.txt
python3 tools/r04/summ.py $O/bench_*.json $O/kt_driverform.json
cat $O/timeline_f32.txt; head -14 $O/kernel_stats_driverform.txt | cut -c1-190
