R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2p; mkdir -p $O
cd $R
run() { name=$1; shift; env "$@" timeout 300 python3 bench.py --no-cpu-baseline --no-extras $EXTRA > $O/$name.json 2> $O/$name.err; python3 - <<PY
import json
try:
    d=json.loads([l for l in open("$O/$name.json") if l.startswith("{")][-1])
    print("$name", d["value"], "p50", d["latency_ms"]["p50_model_chunk"], "p95", d["latency_ms"]["p95_model_chunk"], "iters", d["stage_ms_per_model_step"]["decode_iters"], "cell", d["roofline"]["launch_us"], "tok", d["tokens_per_frame"])
except Exception as e: print("$name ERR", e)
PY
}
run pe2 A=1
run pe1 LASR_PE_MT=1
run pe2b A=1
run pe1b LASR_PE_MT=1
EXTRA="--dtype bf16" run bf_pe2 A=1
EXTRA="--dtype bf16" run bf_pe1 LASR_PE_MT=1
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
