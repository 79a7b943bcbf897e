R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3e; mkdir -p $O
cd $R
B="timeout 120 python3 bench.py --no-cpu-baseline --check-rows 0 --no-extras"
run() { name=$1; shift; "$@" > $O/$name.json 2> $O/$name.err; python3 - <<PY
import json
try:
    d=json.loads([l for l in open("$O/$name.json") if l.startswith("{")][-1])
    r=d["roofline"]
    print("$name", d["value"], "p50", d["latency_ms"]["p50_model_chunk"], "iters", d["stage_ms_per_model_step"]["decode_iters"], "cell", r["launch_us"], "ev", r.get("launch_us_events"), "tok/frame", d.get("tokens_per_frame"))
except Exception as e: print("$name ERR", e)
PY
}
run base $B
LASR_DBG_FE_SIDE=1 run fe_side $B
LASR_DBG_FE_SIDE=1 run fe_side2 $B
run base2 $B
run bf16 $B --dtype bf16
LASR_DBG_FE_SIDE=1 run bf16_fe_side $B --dtype bf16
