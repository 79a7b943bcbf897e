#!/bin/bash
# kernel stats of configs[4] (cfg5, bf16, beam 8, 128 streams, 6 steps in flight)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4h; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python3 $R/bench.py --gpus 1 --no-cpu-baseline --no-extras --check-rows 0 --model cfg5 --dtype bf16 --streams 128 --beam 8 --steps 8 --warmup 2 --depth 6 > $O/kt_cfg5.json 2>$O/kt_cfg5.err
python3 $R/tools/rocpd_stats.py $O/kt/kt_results.db $O/kernel_stats_cfg5_beam8.txt > /dev/null 2>&1
rm -rf $O/kt
head -24 $O/kernel_stats_cfg5_beam8.txt | cut -c1-200
