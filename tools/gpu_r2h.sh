R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2h; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
run() { name=$1; shift; env "$@" timeout 200 python3 bench.py --no-cpu-baseline --no-extras $EXTRA > $O/$name.json 2> $O/$name.err; python3 - <<PY
import json
try:
    d=json.loads([l for l in open("$O/$name.json") if l.startswith("{")][-1])
    print("$name", d["value"], "p50", d["latency_ms"]["p50_model_chunk"], "p95", d["latency_ms"]["p95_model_chunk"], "iters", d["stage_ms_per_model_step"]["decode_iters"], "cell", d["roofline"]["launch_us"], "tok", d["tokens_per_frame"])
except Exception as e: print("$name ERR", e)
PY
}
EXTRA="--depth 12" run fused_d12 LASR_DEC_PRIO=1
EXTRA="--depth 12" run legacy_d12 LASR_DEC_PRIO=1 LASR_FE_LEGACY=1
EXTRA="--depth 12" run fused_la2_d12 LASR_DEC_PRIO=1 LASR_LOOKAHEAD=2 LASR_LOGITS_MT=2
EXTRA="--depth 12 --dtype bf16" run bf_fused_d12 LASR_DEC_PRIO=1
EXTRA="--depth 12 --dtype bf16" run bf_fused_la2_d12 LASR_DEC_PRIO=1 LASR_LOOKAHEAD=2 LASR_LOGITS_MT=2
LASR_DEC_PRIO=1 timeout 200 python3 bench.py --no-cpu-baseline --no-extras --depth 12 --trace $O/trace.json > /dev/null 2>&1
python3 tools/stream_timeline.py $O/trace.json
