# round-6 evidence run (under gpurun): the driver-form line with its other_configs legs, rocprofv3 kernel stats of the same command and
# of configs[2] / [4] on the FINAL build, PMC passes of the cfg5 tiling-D cell IN THE JOB (beam 8 decode beside it) and isolated,
# stream timeline, the 8-rank host dry run (pump nap default + NUMA pinning).  Summaries -> profiles/r06/ by tools/r06/collect_profiles.py.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6final; mkdir -p $O
cd $R
export TMPDIR=/tmp
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driverform.json 2> $O/bench_driverform.err ) 2> $O/bench_driverform.time
B="timeout 300 python3 bench.py --no-cpu-baseline --no-extras --sustained-s 0 --other-configs 0"
$B --dtype bf16 > $O/bench_bf16.json 2>/dev/null
$B --dtype bf16 --beam 4 --steps 4 --warmup 1 > $O/bench_bf16_beam4.json 2>/dev/null
$B --model cfg5 --dtype bf16 --streams 128 --beam 8 --steps 4 --warmup 1 > $O/bench_cfg5_bf16_beam8.json 2>/dev/null
$B --model cfg5 --dtype bf16 --streams 128 --steps 8 --warmup 2 > $O/bench_cfg5_bf16.json 2>/dev/null
$B --no-pipeline --steps 4 --warmup 1 > $O/bench_f32_sync.json 2>/dev/null
LASR_BENCH_BACKEND=gloo LASR_BENCH_SAME_GPU=1 timeout 600 python3 bench.py --gpus 8 --steps 6 --warmup 2 --no-cpu-baseline --no-extras --sustained-s 0 --other-configs 0 --check-rows 8 > $O/bench_8rank_dryrun.json 2> $O/bench_8rank_dryrun.err; echo rc=$? >> $O/bench_8rank_dryrun.err
LASR_PUMP_NAP_PCT=0 LASR_BENCH_NUMA_PIN=0 LASR_BENCH_BACKEND=gloo LASR_BENCH_SAME_GPU=1 timeout 600 python3 bench.py --gpus 8 --steps 6 --warmup 2 --no-cpu-baseline --no-extras --sustained-s 0 --other-configs 0 --check-rows 8 > $O/bench_8rank_dryrun_nap0.json 2>/dev/null
LASR_BENCH_FORCE_DIST=1 $B --steps 8 > $O/bench_rccl_world1.json 2>/dev/null
$B --steps 4 --trace $O/trace_f32.json > /dev/null 2>&1
python3 tools/stream_timeline.py $O/trace_f32.json > $O/timeline_f32.txt 2>&1
cd /tmp
KT="timeout 500 rocprofv3 --kernel-trace --stats"
$KT -d $O/kt_driver -o kt -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --sustained-s 0 --other-configs 0 > $O/kt_driverform.json 2>$O/kt_driver.err
$KT -d $O/kt_beam -o kt -- python3 $R/bench.py --no-cpu-baseline --no-extras --sustained-s 0 --other-configs 0 --dtype bf16 --beam 4 --steps 4 --warmup 1 > $O/kt_bf16_beam4.json 2>/dev/null
$KT -d $O/kt_cfg5 -o kt -- python3 $R/bench.py --no-cpu-baseline --no-extras --sustained-s 0 --other-configs 0 --model cfg5 --dtype bf16 --streams 128 --beam 8 --steps 4 --warmup 1 > $O/kt_cfg5_bf16_beam8.json 2>/dev/null
$KT -d $O/kt_bf16 -o kt -- python3 $R/bench.py --no-cpu-baseline --no-extras --sustained-s 0 --other-configs 0 --dtype bf16 > $O/kt_bf16.json 2>/dev/null
cd $R
for n in driver beam cfg5 bf16; do
  python3 tools/rocpd_stats.py $O/kt_$n/kt_results.db $O/kernel_stats_$n.txt > /dev/null 2>&1
done
python3 tools/rocpd_gaps.py $O/kt_driver/kt_results.db > $O/kernel_gaps_driver.txt 2>&1
rm -rf $O/kt_driver $O/kt_beam $O/kt_cfg5 $O/kt_bf16 $O/trace_f32.json
# PMC: the cfg5 tiling-D cell IN THE JOB (bench.py, beam 8: 1024 hypothesis rows of decode GEMMs beside it) and alone (cellbench), one group per pass
cd /tmp
n=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  n=$((n+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $O/pmcjob_cfg5_$n -o pmc -- python3 $R/bench.py --no-cpu-baseline --no-extras --sustained-s 0 --other-configs 0 --check-rows 0 --prof-steps 0 --model cfg5 --dtype bf16 --streams 128 --beam 8 --steps 2 --warmup 1 > /dev/null 2>&1
  LASR_DTYPE=bf16 LASR_BENCH_LAYERS=1 timeout 200 rocprofv3 --kernel-trace --pmc $grp -d $O/pmciso_cfg5_$n -o pmc -- python3 $R/tools/cellbench.py cfg5 30 128 > /dev/null 2>&1
done
cd $R
for f in $O/pmcjob_cfg5_*/pmc_results.db $O/pmciso_cfg5_*/pmc_results.db; do echo "== $f"; python3 tools/rocpd_pmc.py $f --filter EpiLSTMe; done > $O/cell_pmc_cfg5_job_vs_isolated.txt 2>&1
rm -rf $O/pmcjob_cfg5_* $O/pmciso_cfg5_*
python3 tools/r05/summ.py $O/bench_*.json $O/kt_*.json 2>/dev/null
cat $O/bench_driverform.time; cat $O/timeline_f32.txt; head -14 $O/kernel_stats_driver.txt | cut -c1-230; head -12 $O/kernel_stats_cfg5.txt | cut -c1-230; head -80 $O/cell_pmc_cfg5_job_vs_isolated.txt
