set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 400 python $R/bench.py > $O/bench_f32.json 2> $O/bench_f32.err
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_f32 -o kt -- python $R/bench.py --no-cpu-baseline > /dev/null 2>&1
timeout 300 python $R/bench.py --no-cpu-baseline --dtype bf16 > $O/bench_bf16.json 2>/dev/null
timeout 300 python $R/bench.py --no-cpu-baseline --dtype bf16 --beam 4 > $O/bench_bf16_beam4.json 2>/dev/null
timeout 300 python $R/bench.py --no-cpu-baseline --host-pcm > $O/bench_f32_hostpcm.json 2>/dev/null
timeout 300 python $R/bench.py --no-cpu-baseline --no-pipeline > $O/bench_f32_sync.json 2>/dev/null
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $grp | tr ' ' '_')
  LASR_BENCH_LAYERS=1 timeout 200 rocprofv3 --kernel-trace --pmc $grp -d $O/pmc_f32_$n -o pmc -- python $R/tools/cellbench.py cfg2 30 > /dev/null 2>&1
  LASR_DTYPE=bf16 LASR_BENCH_LAYERS=1 timeout 200 rocprofv3 --kernel-trace --pmc $grp -d $O/pmc_bf16_$n -o pmc -- python $R/tools/cellbench.py cfg2 30 > /dev/null 2>&1
done
ls -R $O | head -40
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_beam -o kt -- python $R/bench.py --no-cpu-baseline --dtype bf16 --beam 4 --steps 60 --warmup 10 > /dev/null 2>&1
timeout 300 python $R/bench.py --no-cpu-baseline --model cfg5 --dtype bf16 --beam 8 --streams 128 --steps 100 > $O/bench_cfg5_bf16_beam8.json 2>/dev/null
timeout 300 python $R/bench.py --no-cpu-baseline --model cfg5 --dtype bf16 --streams 128 > $O/bench_cfg5_bf16.json 2>/dev/null
