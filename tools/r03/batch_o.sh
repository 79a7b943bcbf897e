R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3o; mkdir -p $O
cd $R
timeout 900 python -u -m pytest tests -m gpu -x -q 2>&1 | tail -6
B="timeout 150 python3 bench.py --no-cpu-baseline --no-extras --steps 40"
run() { name=$1; shift; "$@" > $O/$name.json 2> $O/$name.err; python3 - <<PY
import json
try:
    d=json.loads([l for l in open("$O/$name.json") if l.startswith("{")][-1])
    r=d["roofline"]
    print("$name", d["value"], "p50", d["latency_ms"]["p50_model_chunk"], "iters", d["stage_ms_per_model_step"]["decode_iters"], "cell", r["launch_us"], "ev", r.get("launch_us_events"), "chk", d.get("tokens_checked"), d.get("tokens_equal"), "host", d.get("per_rank",[{}])[0].get("host_us_per_model_step"))
except Exception as e: print("$name ERR", e, open("$O/$name.err").read()[-300:])
PY
}
run graph $B
LASR_MAIN_GRAPH=0 run nograph $B --check-rows 0
run graph2 $B --check-rows 0
LASR_MAIN_GRAPH=0 run nograph2 $B --check-rows 0
run bf16_graph $B --dtype bf16
LASR_MAIN_GRAPH=0 run bf16_nograph $B --dtype bf16 --check-rows 0
run bf16_graph2 $B --dtype bf16 --check-rows 0
LASR_MAIN_GRAPH=0 run bf16_nograph2 $B --dtype bf16 --check-rows 0
