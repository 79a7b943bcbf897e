R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2m; mkdir -p $O
cd $R
run() { name=$1; shift; env "$@" timeout 300 python3 bench.py --no-cpu-baseline --no-extras $EXTRA > $O/$name.json 2> $O/$name.err; python3 - <<PY
import json
try:
    d=json.loads([l for l in open("$O/$name.json") if l.startswith("{")][-1])
    print("$name", d["value"], "p50", d["latency_ms"]["p50_model_chunk"], "p95", d["latency_ms"]["p95_model_chunk"], "iters", d["stage_ms_per_model_step"]["decode_iters"], "cell", d["roofline"]["launch_us"], "tok", d["tokens_per_frame"])
except Exception as e: print("$name ERR", e)
PY
}
run nw1 LASR_DEC_NW4=1
run nw1p0 LASR_DEC_NW4=1 LASR_DEC_PRIO=0
run nw5 LASR_DEC_NW4=5
run nw7 LASR_DEC_NW4=7
run nw1la1 LASR_DEC_NW4=1 LASR_LOOKAHEAD=1
run nw1k4 LASR_DEC_NW4=1 LASR_KICK=4
run nw1k2 LASR_DEC_NW4=1 LASR_KICK=2
EXTRA="--dtype bf16" run bf_base A=1
EXTRA="--dtype bf16" run bf_nw1 LASR_DEC_NW4=1
EXTRA="--dtype bf16" run bf_nw7 LASR_DEC_NW4=7
EXTRA="--dtype bf16" run bf_cellnw4 LASR_CELL_NW=4
EXTRA="--model cfg5 --dtype bf16 --streams 128 --depth 8" run cfg5_base A=1
EXTRA="--model cfg5 --dtype bf16 --streams 128 --depth 8" run cfg5_nw1 LASR_DEC_NW4=1
