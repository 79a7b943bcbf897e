# round-5 profile set (run under gpurun): driver-form bench, rocprofv3 kernel stats of the same command and of configs[2] / [4] on
# the FINAL build, PMC passes of the three encoder-cell kernels (f32 tiling C, bf16 tiling C, cfg5 bf16 tiling D), stream timeline,
# the other BASELINE configs, the 8-rank host dry run.  Summaries are copied to profiles/r05/ by tools/r05/collect_profiles.py.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5final; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driverform.json 2> $O/bench_driverform.err; echo rc=$? >> $O/bench_driverform.err
B="timeout 300 python3 bench.py --no-cpu-baseline --no-extras --sustained-s 0"
$B --steps 100 --warmup 5 > $O/bench_f32_steps100.json 2>/dev/null
$B --dtype bf16 > $O/bench_bf16.json 2>/dev/null
$B --dtype bf16 --beam 4 --steps 10 --warmup 3 > $O/bench_bf16_beam4.json 2>/dev/null
$B --model cfg5 --dtype bf16 --streams 128 --depth 6 > $O/bench_cfg5_bf16.json 2>/dev/null
$B --model cfg5 --dtype bf16 --streams 128 --beam 8 --steps 8 --warmup 2 > $O/bench_cfg5_bf16_beam8.json 2>/dev/null
$B --model cfg5 --dtype bf16 --streams 128 --beam 8 --steps 8 --warmup 2 --depth 6 > $O/bench_cfg5_bf16_beam8_depth6.json 2>/dev/null
$B --no-pipeline > $O/bench_f32_sync.json 2>/dev/null
# 8 ranks on ONE GPU (gloo, no RCCL: it refuses several ranks per device): do eight API + pump + helper thread sets coexist?
LASR_BENCH_BACKEND=gloo LASR_BENCH_SAME_GPU=1 timeout 600 python3 bench.py --gpus 8 --no-cpu-baseline --no-extras --sustained-s 0 --check-rows 8 > $O/bench_8rank_dryrun.json 2> $O/bench_8rank_dryrun.err; echo rc=$? >> $O/bench_8rank_dryrun.err
LASR_BENCH_FORCE_DIST=1 $B --steps 40 > $O/bench_rccl_world1.json 2>/dev/null
$B --trace $O/trace_f32.json > /dev/null 2>&1
python3 tools/stream_timeline.py $O/trace_f32.json > $O/timeline_f32.txt
cd /tmp
KT="timeout 400 rocprofv3 --kernel-trace --stats"
$KT -d $O/kt_driver -o kt -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --sustained-s 0 > $O/kt_driverform.json 2>$O/kt_driver.err
$KT -d $O/kt_steady -o kt -- python3 $R/bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu-baseline --no-extras --sustained-s 0 --check-rows 0 --prof-steps 0 > $O/kt_steady.json 2>/dev/null
$KT -d $O/kt_bf16 -o kt -- python3 $R/bench.py --no-cpu-baseline --no-extras --sustained-s 0 --dtype bf16 > $O/kt_bf16.json 2>/dev/null
$KT -d $O/kt_beam -o kt -- python3 $R/bench.py --no-cpu-baseline --no-extras --sustained-s 0 --dtype bf16 --beam 4 --steps 10 --warmup 3 > $O/kt_bf16_beam4.json 2>/dev/null
$KT -d $O/kt_cfg5 -o kt -- python3 $R/bench.py --no-cpu-baseline --no-extras --sustained-s 0 --model cfg5 --dtype bf16 --streams 128 --beam 8 --steps 8 --warmup 2 > $O/kt_cfg5_bf16_beam8.json 2>/dev/null
$KT -d $O/kt_cfg5g -o kt -- python3 $R/bench.py --no-cpu-baseline --no-extras --sustained-s 0 --model cfg5 --dtype bf16 --streams 128 --depth 6 > $O/kt_cfg5_bf16.json 2>/dev/null
cd $R
python3 tools/rocpd_steady.py $O/kt_steady/kt_results.db > $O/kernel_steady_f32.txt 2>&1
for n in driver steady bf16 beam cfg5 cfg5g; do
  python3 tools/rocpd_stats.py $O/kt_$n/kt_results.db $O/kernel_stats_$n.txt > /dev/null 2>&1
done
python3 tools/rocpd_gaps.py $O/kt_driver/kt_results.db > $O/kernel_gaps_driver.txt 2>&1
rm -rf $O/kt_driver $O/kt_steady $O/kt_bf16 $O/kt_beam $O/kt_cfg5 $O/kt_cfg5g $O/trace_f32.json
# PMC passes on the isolated cells, one counter group per pass (gpurun refuses --pmc with the hip / hsa trace domains)
# (SKIP_PMC=1: the cell kernels are unchanged since the committed passes -- profiles/cell_pmc.json, profiles/r05/r05_cell_pmc.txt)
if [ -z "$SKIP_PMC" ]; then
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "TCC_[A-Z0-9_]*\|TCP_TCC_[A-Z0-9_]*\|SQ_VALU_MFMA[A-Z0-9_]*\|SQ_BUSY[A-Z0-9_]*" | sort -u | head -80 > $O/pmc_counter_names.txt
n=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  n=$((n+1))
  LASR_BENCH_LAYERS=1 timeout 200 rocprofv3 --kernel-trace --pmc $grp -d $O/pmc_f32_$n -o pmc -- python3 $R/tools/cellbench.py cfg2 30 > /dev/null 2>&1
  LASR_DTYPE=bf16 LASR_BENCH_LAYERS=1 timeout 200 rocprofv3 --kernel-trace --pmc $grp -d $O/pmc_bf16_$n -o pmc -- python3 $R/tools/cellbench.py cfg2 30 > /dev/null 2>&1
  LASR_DTYPE=bf16 LASR_BENCH_LAYERS=1 timeout 200 rocprofv3 --kernel-trace --pmc $grp -d $O/pmc_cfg5_$n -o pmc -- python3 $R/tools/cellbench.py cfg5 30 128 > /dev/null 2>&1
done
cd $R
for f in $O/pmc_*/pmc_results.db; do echo "== $f"; python3 tools/rocpd_pmc.py $f --filter EpiLSTM; done > $O/cell_pmc.txt 2>&1
rm -rf $O/pmc_f32_* $O/pmc_bf16_* $O/pmc_cfg5_*
fi
python3 tools/r05/summ.py $O/bench_*.json $O/kt_*.json
cat $O/timeline_f32.txt; head -14 $O/kernel_stats_driver.txt | cut -c1-190; [ -f $O/cell_pmc.txt ] && head -60 $O/cell_pmc.txt; true
