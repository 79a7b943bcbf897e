R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2k; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -5
run() { name=$1; shift; env "$@" timeout 300 python3 bench.py --no-cpu-baseline --no-extras $EXTRA > $O/$name.json 2> $O/$name.err; python3 - <<PY
import json
try:
    d=json.loads([l for l in open("$O/$name.json") if l.startswith("{")][-1])
    print("$name", d["value"], "p50", d["latency_ms"]["p50_model_chunk"], "p95", d["latency_ms"]["p95_model_chunk"], "iters", d["stage_ms_per_model_step"]["decode_iters"], "cell", d["roofline"]["launch_us"], "tok", d["tokens_per_frame"])
except Exception as e: print("$name ERR", e)
PY
}
run f32 A=1
run f32b A=1
run f32_k2 LASR_KICK=2
run f32_k4 LASR_KICK=4
run f32_g2 LASR_GROUP=2
EXTRA="--dtype bf16" run bf16 A=1
timeout 200 python3 bench.py --no-cpu-baseline --no-extras --trace $O/trace.json > /dev/null 2>&1
python3 tools/stream_timeline.py $O/trace.json
