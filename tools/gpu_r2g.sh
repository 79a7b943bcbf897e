R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2g; mkdir -p $O
cd $R
run() { name=$1; shift; env "$@" timeout 200 python3 bench.py --no-cpu-baseline --no-extras $EXTRA > $O/$name.json 2> $O/$name.err; python3 - <<PY
import json
try:
    d=json.loads([l for l in open("$O/$name.json") if l.startswith("{")][-1])
    print("$name", d["value"], "p50", d["latency_ms"]["p50_model_chunk"], "p95", d["latency_ms"]["p95_model_chunk"], "iters", d["stage_ms_per_model_step"]["decode_iters"], "cell", d["roofline"]["launch_us"], "tok", d["tokens_per_frame"])
except Exception as e: print("$name ERR", e)
PY
}
for d in 6 8 10 12 15; do EXTRA="--depth $d" run d$d LASR_DEC_PRIO=1; done
for d in 8 12 15; do EXTRA="--depth $d" run la2_d$d LASR_DEC_PRIO=1 LASR_LOOKAHEAD=2 LASR_LOGITS_MT=2; done
EXTRA="--depth 12" run la3_d12 LASR_DEC_PRIO=1 LASR_LOOKAHEAD=3 LASR_LOGITS_MT=2
EXTRA="--depth 12 --dtype bf16" run bf_d12 LASR_DEC_PRIO=1
EXTRA="--depth 12 --dtype bf16" run bf_la2_d12 LASR_DEC_PRIO=1 LASR_LOOKAHEAD=2 LASR_LOGITS_MT=2
LASR_DEC_PRIO=1 LASR_LOOKAHEAD=2 LASR_LOGITS_MT=2 timeout 200 python3 bench.py --no-cpu-baseline --no-extras --depth 12 --trace $O/trace_la2.json > /dev/null 2>&1
python3 tools/stream_timeline.py $O/trace_la2.json
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
