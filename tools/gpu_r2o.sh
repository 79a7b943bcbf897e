R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2o; mkdir -p $O
cd $R
timeout 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo rc=$? >> $O/bench_driver.err
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_driver -o kt -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/kt_driver.json 2>$O/kt_driver.err
rocprofv3 -L > $O/counters.txt 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  n=$(echo $grp | tr ' ' '_' | cut -c1-40)
  LASR_BENCH_LAYERS=1 timeout 200 rocprofv3 --kernel-trace --pmc $grp -d $O/pmc_f32_$n -o pmc -- python3 $R/tools/cellbench.py cfg2 30 > $O/pmc_f32_$n.log 2>&1
  LASR_DTYPE=bf16 LASR_BENCH_LAYERS=1 timeout 200 rocprofv3 --kernel-trace --pmc $grp -d $O/pmc_bf16_$n -o pmc -- python3 $R/tools/cellbench.py cfg2 30 > $O/pmc_bf16_$n.log 2>&1
done
cd $R
python3 tools/cellbench.py cfg2 300 > $O/cellbench_f32.txt 2>&1
LASR_DTYPE=bf16 python3 tools/cellbench.py cfg2 300 > $O/cellbench_bf16.txt 2>&1
ls $O
cat $O/cellbench_f32.txt $O/cellbench_bf16.txt
