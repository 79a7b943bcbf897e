R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3m; mkdir -p $O
cd $R
B="timeout 150 python3 bench.py --no-cpu-baseline --check-rows 0 --no-extras --steps 40"
run() { name=$1; shift; "$@" > $O/$name.json 2> $O/$name.err; python3 - <<PY
import json
try:
    d=json.loads([l for l in open("$O/$name.json") if l.startswith("{")][-1])
    r=d["roofline"]
    print("$name", d["value"], "p50", d["latency_ms"]["p50_model_chunk"], "iters", d["stage_ms_per_model_step"]["decode_iters"], "cell", r["launch_us"], "host", d.get("per_rank",[{}])[0].get("host_us_per_model_step"))
except Exception as e: print("$name ERR", e, open("$O/$name.err").read()[-300:])
PY
}
run base $B
LASR_NO_GRAPH=1 run nograph $B
run base2 $B
LASR_NO_GRAPH=1 run nograph2 $B
LASR_NO_GRAPH=1 LASR_GROUP=2 run nograph_g2 $B
run bf16 $B --dtype bf16
LASR_NO_GRAPH=1 run bf16_nograph $B --dtype bf16
