# round-3 profile set (run under gpurun): GPU tests, driver-form bench + rocprofv3 kernel stats of the same command, the other
# BASELINE configs on both protocols, stream timeline, 2-rank dry run, smoke.  Summaries are copied to profiles/r03/ by hand.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3final; mkdir -p $O
cd $R
rm -f gpurun_out/parity_counts.json gpurun_out/served_rate.json
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
cp gpurun_out/parity_counts.json gpurun_out/served_rate.json $O/ 2>/dev/null
timeout 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driverform.json 2> $O/bench_driverform.err; echo rc=$? >> $O/bench_driverform.err
B="timeout 300 python3 bench.py --no-cpu-baseline --no-extras --check-rows 0"
$B --steps 100 --warmup 5 > $O/bench_f32_steps100.json 2>/dev/null
$B --dtype bf16 > $O/bench_bf16.json 2>/dev/null
$B --dtype bf16 --beam 4 --steps 10 --warmup 3 > $O/bench_bf16_beam4.json 2>/dev/null
$B --dtype bf16 --beam 4 --steps 10 --warmup 3 --depth 12 > $O/bench_bf16_beam4_depth12.json 2>/dev/null
$B --dtype bf16 --beam 4 --steps 10 --warmup 3 --no-pipeline > $O/bench_bf16_beam4_sync.json 2>/dev/null
$B --model cfg5 --dtype bf16 --streams 128 --depth 6 > $O/bench_cfg5_bf16.json 2>/dev/null
$B --model cfg5 --dtype bf16 --streams 128 --beam 8 --steps 8 --warmup 2 --depth 6 > $O/bench_cfg5_bf16_beam8.json 2>/dev/null
$B --model cfg5 --dtype bf16 --streams 128 --beam 8 --steps 8 --warmup 2 --depth 3 > $O/bench_cfg5_bf16_beam8_depth3.json 2>/dev/null
$B --model cfg5 --dtype bf16 --streams 128 --beam 8 --steps 8 --warmup 2 --no-pipeline > $O/bench_cfg5_bf16_beam8_sync.json 2>/dev/null
$B --no-pipeline > $O/bench_f32_sync.json 2>/dev/null
$B --depth 6 > $O/bench_f32_depth6.json 2>/dev/null
LASR_BENCH_BACKEND=gloo LASR_BENCH_SAME_GPU=1 timeout 300 python3 bench.py --gpus 2 --no-cpu-baseline --no-extras > $O/bench_2rank_dryrun.json 2> $O/bench_2rank_dryrun.err; echo rc=$? >> $O/bench_2rank_dryrun.err
$B --trace $O/trace_f32.json > /dev/null 2>&1
python3 tools/stream_timeline.py $O/trace_f32.json > $O/timeline_f32.txt
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_driver -o kt -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --check-rows 0 > $O/kt_driverform.json 2>$O/kt_driver.err
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_beam -o kt -- python3 $R/bench.py --no-cpu-baseline --no-extras --check-rows 0 --dtype bf16 --beam 4 --steps 6 --warmup 2 > /dev/null 2>&1
cd $R
python3 tools/rocpd_stats.py $O/kt_driver/kt_results.db $O/kernel_stats_driverform.txt > /dev/null 2>&1
python3 tools/rocpd_gaps.py $O/kt_driver/kt_results.db > $O/kernel_gaps_driverform.txt 2>&1
python3 tools/rocpd_stats.py $O/kt_beam/kt_results.db $O/kernel_stats_bf16_beam4.txt > /dev/null 2>&1
rm -rf $O/kt_driver $O/kt_beam $O/trace_f32.json
for f in bench_driverform bench_f32_steps100 bench_bf16 bench_bf16_beam4 bench_bf16_beam4_depth12 bench_bf16_beam4_sync bench_cfg5_bf16 bench_cfg5_bf16_beam8 bench_cfg5_bf16_beam8_depth3 bench_cfg5_bf16_beam8_sync bench_f32_sync bench_f32_depth6 bench_2rank_dryrun kt_driverform; do python3 - <<PY
import json
try:
    d=json.loads([l for l in open("$O/$f.json") if l.startswith("{")][-1])
    print("$f", d["value"], "n_gpus", d["n_gpus"], "p50", d["latency_ms"]["p50_model_chunk"], "p95", d["latency_ms"]["p95_model_chunk"], "iters", d["stage_ms_per_model_step"]["decode_iters"], "cell", d["roofline"]["launch_us"], "frac", d["roofline"]["frac"], "pcie", d.get("pcie_inclusive",{}).get("value"), d.get("pcie_inclusive",{}).get("pinned_nocopy",{}).get("value"), "cpu", d.get("cpu_baseline",{}).get("value"), d.get("cpu_baseline",{}).get("best_effort",{}).get("value"), "chk", d.get("tokens_checked"), d.get("tokens_equal"))
except Exception as e: print("$f ERR", e)
PY
done
cat $O/timeline_f32.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
# PMC passes on the isolated cell (one counter group per pass; gpurun refuses --pmc together with the hip / hsa trace domains)
if [ -n "$LASR_PROFILE_PMC" ]; then
  cd /tmp
  for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
    n=$(echo $grp | tr ' ' '_' | cut -c1-40)
    LASR_BENCH_LAYERS=1 timeout 200 rocprofv3 --kernel-trace --pmc $grp -d $O/pmc_f32_$n -o pmc -- python3 $R/tools/cellbench.py cfg2 30 > /dev/null 2>&1
    LASR_DTYPE=bf16 LASR_BENCH_LAYERS=1 timeout 200 rocprofv3 --kernel-trace --pmc $grp -d $O/pmc_bf16_$n -o pmc -- python3 $R/tools/cellbench.py cfg2 30 > /dev/null 2>&1
  done
  cd $R
  for f in $O/pmc_*/pmc_results.db; do echo "== $f"; python3 tools/rocpd_pmc.py $f --filter EpiLSTM; done > $O/cell_pmc.txt 2>&1
  rm -rf $O/pmc_*
  cat $O/cell_pmc.txt
fi
