# round 3, GPU batch C: new tests, split front-end + push_submit A/B
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_round3.py -x -q > $O/pytest_r3.log 2>&1; tail -15 $O/pytest_r3.log
timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_round3.py > $O/pytest.log 2>&1; tail -5 $O/pytest.log
B="timeout 200 python3 bench.py --no-cpu-baseline --check-rows 0"
run() { name=$1; shift; "$@" > $O/$name.json 2> $O/$name.err; python3 - <<PY
import json
try:
    d=json.loads([l for l in open("$O/$name.json") if l.startswith("{")][-1])
    r=d["roofline"]
    print("$name", d["value"], "p50", d["latency_ms"]["p50_model_chunk"], "iters", d["stage_ms_per_model_step"]["decode_iters"], "cell", r["launch_us"], "ev", r.get("launch_us_events"), "pcie", d.get("pcie_inclusive",{}).get("value"), d.get("pcie_inclusive",{}).get("pinned_nocopy",{}).get("value"), "host", d.get("per_rank",[{}])[0].get("host_us_per_model_step"))
except Exception as e: print("$name ERR", e)
PY
}
run split_fused $B --no-extras
run split_unfused $B --no-extras --split-push
LASR_FE_MODE=0 run onelaunch_fused $B --no-extras
LASR_FE_MODE=0 run onelaunch_unfused $B --no-extras --split-push
run split_fused2 $B --no-extras
run with_extras $B
LASR_PUSH_THREADS=0 run with_extras_nothreads $B
LASR_PUSH_THREADS=4 run with_extras_4threads $B
run bf16 $B --no-extras --dtype bf16
LASR_FE_MODE=0 run bf16_onelaunch $B --no-extras --dtype bf16 --split-push
run full_check python3 bench.py --gpus 1 --steps 20 --warmup 5
$B --no-extras --trace $O/trace_f32.json > /dev/null 2>&1
python3 tools/stream_timeline.py $O/trace_f32.json > $O/timeline_f32.txt; cat $O/timeline_f32.txt
