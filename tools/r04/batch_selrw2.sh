R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4selrw2; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_round4.py -x -q -k "one_wave_per_row" 2>&1 | tail -2
timeout 600 python -m pytest tests/test_gpu_beam.py tests/test_gpu_soak.py -x -q 2>&1 | tail -2
B="timeout 300 python3 bench.py --no-cpu-baseline --no-extras --check-rows 0"
for i in 1 2; do
  $B --dtype bf16 --beam 4 --steps 10 --warmup 3 > $O/cfg2b4_$i.json 2>/dev/null
  $B --model cfg5 --dtype bf16 --streams 128 --beam 8 --steps 8 --warmup 2 > $O/cfg5b8_$i.json 2>/dev/null
done
python3 tools/r04/summ.py $O/*.json
timeout 200 python3 tools/r04/select_phases.py cfg5 8 128 2>/dev/null | tail -1; timeout 200 python3 tools/r04/select_phases.py cfg2 4 64 2>/dev/null | tail -1
