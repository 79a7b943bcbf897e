# round 3, GPU batch A: tests, A/B of the round-2 library, cost of in-region timers, CU-mask sweep, kernel-trace gaps
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3a; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
B="timeout 200 python3 bench.py --no-cpu-baseline --no-extras"
run() { name=$1; shift; "$@" > $O/$name.json 2> $O/$name.err; python3 - <<PY
import json
try:
    d=json.loads([l for l in open("$O/$name.json") if l.startswith("{")][-1])
    r=d["roofline"]
    print("$name", d["value"], "p50", d["latency_ms"]["p50_model_chunk"], "iters", d["stage_ms_per_model_step"]["decode_iters"], "cell", r["launch_us"], "frac", r["frac"], "prof", r.get("value_profiled"), "chk", d.get("tokens_checked"), d.get("tokens_equal"), "host", d.get("per_rank",[{}])[0].get("host_us_per_model_step"))
except Exception as e: print("$name ERR", e)
PY
}
LASR_LIB=$R/libreasr_amd/csrc/liblasr_base_r02.so run base_r02 $B --prof-steps 0 --check-rows 0
LASR_LIB=$R/libreasr_amd/csrc/liblasr_base_r02.so run base_r02_prof1 $B --cell-prof-in-timed 1 --check-rows 0
run new $B
run new_prof1 $B --cell-prof-in-timed 1 --check-rows 0
run new_prof2 $B --cell-prof-in-timed 2 --check-rows 0
run new_again $B --prof-steps 0 --check-rows 0
for dc in 32 64 96 128; do LASR_DEC_CUS=$dc run mask_dec$dc $B --check-rows 0; done
LASR_MAIN_CUS=256 run mask_main256 $B --check-rows 0
LASR_DEC_CUS=64 LASR_MAIN_CUS=192 run mask_dec64_main192 $B --check-rows 0
LASR_DEC_CUS=32 LASR_MAIN_CUS=224 run mask_dec32_main224 $B --check-rows 0
LASR_DEC_CUS=64 LASR_MAIN_CUS=256 run mask_dec64_main256 $B --check-rows 0
LASR_DEC_CUS=128 LASR_DEC_CU_OFF=128 run mask_dec128hi $B --check-rows 0
run new_bf16 $B --dtype bf16
run new_steps200 $B --steps 100 --warmup 5 --prof-steps 0 --check-rows 0
$B --prof-steps 0 --check-rows 0 --trace $O/trace_f32.json > /dev/null 2>&1
python3 tools/stream_timeline.py $O/trace_f32.json > $O/timeline_f32.txt; cat $O/timeline_f32.txt
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --prof-steps 0 --check-rows 0 > $O/kt_bench.json 2> $O/kt.err
cd $R
python3 tools/rocpd_gaps.py $O/kt/kt_results.db > $O/kt_gaps.txt 2>&1; cat $O/kt_gaps.txt
python3 tools/rocpd_stats.py $O/kt/kt_results.db $O/kt_stats.txt > /dev/null 2>&1; head -20 $O/kt_stats.txt
rm -rf $O/kt/*.db   # keep the summaries only (size)
