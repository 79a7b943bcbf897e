R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2l; mkdir -p $O
cd $R
run() { name=$1; shift; env "$@" timeout 300 python3 bench.py --no-cpu-baseline --no-extras $EXTRA > $O/$name.json 2> $O/$name.err; python3 - <<PY
import json
try:
    d=json.loads([l for l in open("$O/$name.json") if l.startswith("{")][-1])
    print("$name", d["value"], "p50", d["latency_ms"]["p50_model_chunk"], "p95", d["latency_ms"]["p95_model_chunk"], "iters", d["stage_ms_per_model_step"]["decode_iters"], "cell", d["roofline"]["launch_us"], "tok", d["tokens_per_frame"])
except Exception as e: print("$name ERR", e)
PY
}
run base A=1
run nw1 LASR_DEC_NW4=1
run nw2 LASR_DEC_NW4=2
run nw3 LASR_DEC_NW4=3
run prio0 LASR_DEC_PRIO=0
run prio2 LASR_DEC_PRIO=2
run prio3 LASR_DEC_PRIO=3
run cellnw8 LASR_CELL_NW=8
run la1 LASR_LOOKAHEAD=1
run la1mt1 LASR_LOOKAHEAD=1 LASR_LOGITS_MT=1
EXTRA="--depth 15" run d15 A=1
EXTRA="--depth 9" run d9 A=1
