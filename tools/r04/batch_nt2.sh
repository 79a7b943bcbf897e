R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4nt2; mkdir -p $O; cd $R
B="timeout 300 python3 bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 5"
for i in 1 2 3; do for v in 0 1; do
  LASR_LOGITS_NT2=$v $B > $O/f32_nt2_${v}_$i.json 2>/dev/null
done; done
for v in 0 1; do LASR_LOGITS_NT2=$v $B --dtype bf16 > $O/bf16_nt2_${v}.json 2>/dev/null; done
python3 tools/r04/summ.py $O/*.json
