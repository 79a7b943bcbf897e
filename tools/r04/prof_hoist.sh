#!/bin/bash
# kernel stats of the driver-form bench with and without the hoisted predictor
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4e; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for h in 0 1; do
  LASR_HOIST=$h timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_h$h -o kt -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --check-rows 0 > $O/kt_h$h.json 2>$O/kt_h$h.err
  python3 $R/tools/rocpd_stats.py $O/kt_h$h/kt_results.db $O/kernel_stats_hoist$h.txt > /dev/null 2>&1
  rm -rf $O/kt_h$h
done
cd $R
for h in 0 1; do
  LASR_HOIST=$h python3 bench.py --no-cpu-baseline --no-extras --check-rows 0 --trace $O/trace_h$h.json > /dev/null 2>&1
  python3 tools/stream_timeline.py $O/trace_h$h.json > $O/timeline_h$h.txt; rm -f $O/trace_h$h.json
done
head -14 $O/kernel_stats_hoist0.txt | cut -c1-190; head -14 $O/kernel_stats_hoist1.txt | cut -c1-190; cat $O/timeline_h0.txt $O/timeline_h1.txt
