R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4pair; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python3 $R/bench.py --no-cpu-baseline --no-extras --check-rows 0 --lm fp32 --steps 60 --warmup 10 > $O/kt_pair.json 2>$O/kt.err
(cd $R; python3 tools/rocpd_stats.py $O/kt/kt_results.db $O/kernel_stats_lm_pair.txt > /dev/null 2>&1)
rm -rf $O/kt
head -22 $O/kernel_stats_lm_pair.txt | cut -c1-250
