# round-2 profile set (run under gpurun): driver-form bench + rocprofv3 kernel stats of the same command + the extra lines
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2final; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driverform.json 2> $O/bench_driverform.err; echo rc=$? >> $O/bench_driverform.err
timeout 300 python3 bench.py --no-cpu-baseline --dtype bf16 > $O/bench_bf16.json 2>/dev/null
timeout 300 python3 bench.py --no-cpu-baseline --no-extras --dtype bf16 --beam 4 > $O/bench_bf16_beam4.json 2>/dev/null
timeout 300 python3 bench.py --no-cpu-baseline --no-extras --model cfg5 --dtype bf16 --streams 128 --depth 6 > $O/bench_cfg5_bf16.json 2>/dev/null
timeout 300 python3 bench.py --no-cpu-baseline --no-extras --model cfg5 --dtype bf16 --streams 128 --beam 8 --steps 8 --warmup 2 > $O/bench_cfg5_bf16_beam8.json 2>/dev/null
timeout 300 python3 bench.py --no-cpu-baseline --no-extras --no-pipeline > $O/bench_f32_sync.json 2>/dev/null
timeout 300 python3 bench.py --no-cpu-baseline --no-extras --depth 6 > $O/bench_f32_depth6.json 2>/dev/null
LASR_BENCH_BACKEND=gloo LASR_BENCH_SAME_GPU=1 timeout 300 python3 bench.py --gpus 2 --no-cpu-baseline --no-extras > $O/bench_2rank_dryrun.json 2> $O/bench_2rank_dryrun.err; echo rc=$? >> $O/bench_2rank_dryrun.err
timeout 200 python3 bench.py --no-cpu-baseline --no-extras --trace $O/trace_f32.json > /dev/null 2>&1
python3 tools/stream_timeline.py $O/trace_f32.json > $O/timeline_f32.txt
timeout 200 python3 bench.py --no-cpu-baseline --no-extras --dtype bf16 --trace $O/trace_bf16.json > /dev/null 2>&1
python3 tools/stream_timeline.py $O/trace_bf16.json > $O/timeline_bf16.txt
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_driver -o kt -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/kt_driverform.json 2>$O/kt_driver.err
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_beam -o kt -- python3 $R/bench.py --no-cpu-baseline --no-extras --dtype bf16 --beam 4 --steps 4 --warmup 1 > /dev/null 2>&1
cd $R
for f in bench_driverform bench_bf16 bench_bf16_beam4 bench_cfg5_bf16 bench_cfg5_bf16_beam8 bench_f32_sync bench_f32_depth6 bench_2rank_dryrun kt_driverform; do python3 - <<PY
import json
try:
    d=json.loads([l for l in open("$O/$f.json") if l.startswith("{")][-1])
    print("$f", d["value"], "n_gpus", d["n_gpus"], "p50", d["latency_ms"]["p50_model_chunk"], "p95", d["latency_ms"]["p95_model_chunk"], "iters", d["stage_ms_per_model_step"]["decode_iters"], "cell", d["roofline"]["launch_us"], "frac", d["roofline"]["frac"], "pcie", d.get("pcie_inclusive",{}).get("value"), "cpu", d.get("cpu_baseline",{}).get("value"))
except Exception as e: print("$f ERR", e)
PY
done
cat $O/timeline_f32.txt $O/timeline_bf16.txt
# PMC passes on the isolated cell (one counter group per pass; gpurun refuses --pmc together with the hip / hsa trace domains)
if [ -n "$LASR_PROFILE_PMC" ]; then
  cd /tmp
  for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
    n=$(echo $grp | tr ' ' '_' | cut -c1-40)
    LASR_BENCH_LAYERS=1 timeout 200 rocprofv3 --kernel-trace --pmc $grp -d $O/pmc_f32_$n -o pmc -- python3 $R/tools/cellbench.py cfg2 30 > /dev/null 2>&1
    LASR_DTYPE=bf16 LASR_BENCH_LAYERS=1 timeout 200 rocprofv3 --kernel-trace --pmc $grp -d $O/pmc_bf16_$n -o pmc -- python3 $R/tools/cellbench.py cfg2 30 > /dev/null 2>&1
  done
  cd $R
  for f in $O/pmc_*/pmc_results.db; do python3 tools/rocpd_pmc.py $f --filter EpiLSTM; done
fi
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
