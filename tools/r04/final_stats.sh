# kernel stats of the final build (driver form, configs[2], configs[4]) + soak on the final library
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4fin2; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_driver -o kt -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --check-rows 0 > $O/kt_driverform.json 2>$O/kt_driver.err
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_beam -o kt -- python3 $R/bench.py --no-cpu-baseline --no-extras --check-rows 0 --dtype bf16 --beam 4 --steps 6 --warmup 2 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_cfg5 -o kt -- python3 $R/bench.py --no-cpu-baseline --no-extras --check-rows 0 --model cfg5 --dtype bf16 --streams 128 --beam 8 --steps 8 --warmup 2 --depth 6 > /dev/null 2>&1
cd $R
python3 tools/rocpd_stats.py $O/kt_driver/kt_results.db $O/kernel_stats_driverform.txt > /dev/null 2>&1
python3 tools/rocpd_gaps.py $O/kt_driver/kt_results.db > $O/kernel_gaps_driverform.txt 2>&1
python3 tools/rocpd_stats.py $O/kt_beam/kt_results.db $O/kernel_stats_bf16_beam4.txt > /dev/null 2>&1
python3 tools/rocpd_stats.py $O/kt_cfg5/kt_results.db $O/kernel_stats_cfg5_bf16_beam8.txt > /dev/null 2>&1
rm -rf $O/kt_driver $O/kt_beam $O/kt_cfg5
(timeout 300 python tests/soak.py --preamble full --scenario both --iters 1000 --no-dump --out $O/soak_summary.jsonl 2>&1 | tail -2) > $O/soak.txt
cat $O/soak.txt | cut -c1-400; head -9 $O/kernel_stats_bf16_beam4.txt | cut -c1-170; head -9 $O/kernel_stats_cfg5_bf16_beam8.txt | cut -c1-170
