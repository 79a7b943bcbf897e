R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4pair; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_lm.py tests/test_gpu_round4.py -x -q 2>&1 | tail -2
B="timeout 300 python3 bench.py --no-cpu-baseline --no-extras --steps 300 --warmup 50"
for i in 1 2; do for m in 1 0; do for dt in f32 bf16; do
  LASR_LM_PAIR=$m $B --lm fp32 --dtype $dt > $O/lm_${dt}_pair${m}_$i.json 2>/dev/null
done; done; done
python3 tools/r04/summ.py $O/*.json
