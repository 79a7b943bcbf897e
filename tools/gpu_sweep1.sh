R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sw1; mkdir -p $O
cd $R
run() { name=$1; shift; env "$@" timeout 200 python3 bench.py --no-cpu-baseline --no-extras > $O/$name.json 2> $O/$name.err; python3 - <<PY
import json
try:
    d=json.loads([l for l in open("$O/$name.json") if l.startswith("{")][-1])
    print("$name", d["value"], "p50", d["latency_ms"]["p50_model_chunk"], "iters", d["stage_ms_per_model_step"]["decode_iters"], "cell", d["roofline"]["launch_us"])
except Exception as e: print("$name ERR", e)
PY
}
run base A=1
run la2 LASR_LOOKAHEAD=2
run la3 LASR_LOOKAHEAD=3
run la2_k2 LASR_LOOKAHEAD=2 LASR_KICK=2
run la2_k4 LASR_LOOKAHEAD=2 LASR_KICK=4
run nograph LASR_NO_GRAPH=1
run base2 A=1
