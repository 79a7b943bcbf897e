# kernel stats of configs[1] + LM shallow fusion (fp32 / int8-served), LM branch on its own stream and in line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4lmprof; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for lm in fp32 int8; do for side in 1 0; do
  LASR_LM_SIDE=$side timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python3 $R/bench.py --no-cpu-baseline --no-extras --check-rows 0 --lm $lm --steps 60 --warmup 10 > $O/kt_${lm}_side$side.json 2>$O/kt.err
  (cd $R; python3 tools/rocpd_stats.py $O/kt/kt_results.db $O/kernel_stats_lm_${lm}_side$side.txt > /dev/null 2>&1; python3 tools/rocpd_gaps.py $O/kt/kt_results.db > $O/kernel_gaps_lm_${lm}_side$side.txt 2>&1)
  rm -rf $O/kt
done; done
head -30 $O/kernel_stats_lm_fp32_side1.txt
