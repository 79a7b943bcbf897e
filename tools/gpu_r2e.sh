R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2e; mkdir -p $O
cd $R
run() { name=$1; shift; env "$@" timeout 200 python3 bench.py --no-cpu-baseline --no-extras $EXTRA > $O/$name.json 2> $O/$name.err; python3 - <<PY
import json
try:
    d=json.loads([l for l in open("$O/$name.json") if l.startswith("{")][-1])
    print("$name", d["value"], "p50", d["latency_ms"]["p50_model_chunk"], "p95", d["latency_ms"]["p95_model_chunk"], "iters", d["stage_ms_per_model_step"]["decode_iters"], "cell", d["roofline"]["launch_us"], "tok", d["tokens_per_frame"])
except Exception as e: print("$name ERR", e)
PY
}
run base A=1
run mt2 LASR_LOGITS_MT=2
run mt4 LASR_LOGITS_MT=4
run la2 LASR_LOOKAHEAD=2
run la2mt2 LASR_LOOKAHEAD=2 LASR_LOGITS_MT=2
run la2mt4 LASR_LOOKAHEAD=2 LASR_LOGITS_MT=4
run la2mt2k2 LASR_LOOKAHEAD=2 LASR_LOGITS_MT=2 LASR_KICK=2
run la3mt2 LASR_LOOKAHEAD=3 LASR_LOGITS_MT=2
run la4mt4 LASR_LOOKAHEAD=4 LASR_LOGITS_MT=4
EXTRA="--dtype bf16" run bf_la2mt2 LASR_LOOKAHEAD=2 LASR_LOGITS_MT=2
EXTRA="--dtype bf16" run bf_la2mt4 LASR_LOOKAHEAD=2 LASR_LOGITS_MT=4
LASR_LOOKAHEAD=2 LASR_LOGITS_MT=2 timeout 200 python3 bench.py --no-cpu-baseline --no-extras --trace $O/trace_la2.json > /dev/null 2>&1
python3 tools/stream_timeline.py $O/trace_la2.json
LASR_LOOKAHEAD=2 LASR_LOGITS_MT=2 timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
